"""Host-side SETUP of the auxiliary-space preconditioner (the apply runs on the GPU, csrc/amg.hip).

The reference preconditions both Krylov solves with hypre BoomerAMG built from the assembled matrix
(reference: src/knpemidg/solver.py:433, 505, 688, 767).  There is no assembled DG matrix here, so the
preconditioner is the classical two-level splitting of an interior-penalty DG space,

    M^-1  =  B^-1  +  P  Ac^+  P^T ,

* B    cell-block-Jacobi of the DG operator (device, csrc/apply_p1.hip),
* P    injection of the CONFORMING P1 space, broken only across membrane facets (phi and c jump there),
       into DG-P1:  DG dof (cell c, vertex a)  ->  conforming dof (vertex id, membrane-connected component of c),
* Ac   = P^T A P.  Conforming functions have no jumps on ordinary facets, so every SIPG facet term
       vanishes and Ac is the plain P1 stiffness matrix  sum_cells vol * mean(coef) * G  plus the membrane
       coupling  C int_F (u_i - u_e)(v_i - v_e)  (and the mass / dt term for KNP) -- assembled here with numpy,
* Ac^+ one V-cycle of a smoothed-aggregation AMG hierarchy built here (scipy.sparse), smoothed on the
       device with Chebyshev-Jacobi polynomials (no sequential sweeps), coarsest level as a dense
       pseudo-inverse.

Setup is host work done once per solver (the hierarchy is reused across time steps: kappa changes by
<1 % per step and a lagged SPD preconditioner does not change the converged solution).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.csgraph as csg


def _cell_gram(space):
    """(vol [nc], G [nc, nv, nv] = grad lambda_a . grad lambda_b) of every cell, computed once per space."""
    geo = getattr(space, "_gram", None)
    import os
    if geo is None and space.mesh.cells.shape[0] >= 20000 and os.environ.get("KNP_SETUP_NATIVE_GRAM", "1") != "0":
        # threaded closed form in the library (csrc/host_sparse.cpp) instead of batched numpy inverses: 0.6 s -> 0.03 s at 10^6 tets
        from knpemidg import _abi
        mesh = space.mesh
        nc, nv = mesh.cells.shape
        coords = np.ascontiguousarray(mesh.coords, dtype=np.float64)
        cells = np.ascontiguousarray(mesh.cells, dtype=np.int32)
        vol, G = np.empty(nc), np.empty((nc, nv, nv))
        rc = _abi.load().knp_host_cell_gram(nc, mesh.gdim, _abi._p(coords, _abi._f64p), _abi._p(cells, _abi._i32p), _abi._p(vol, _abi._f64p),
                                            _abi._p(G, _abi._f64p), _setup_threads())
        if rc == 0:
            geo = space._gram = (vol, G)
    if geo is None:
        mesh = space.mesh
        d = mesh.gdim
        x = mesh.coords[mesh.cells]
        J = (x[:, 1:, :] - x[:, :1, :]).transpose(0, 2, 1)
        vol = np.abs(np.linalg.det(J)) / {2: 2.0, 3: 6.0}[d]
        Jinv = np.linalg.inv(J)
        g = np.empty((x.shape[0], d + 1, d))
        g[:, 1:, :] = Jinv
        g[:, 0, :] = -Jinv.sum(axis=1)
        geo = space._gram = (vol, np.einsum("cad,cbd->cab", g, g))
    return geo


def _assemble_cached(space, blk, nd):
    """CSR matrix from per-cell dense blocks blk[c, a, b] scattered through space.dof.  The sparsity pattern (sort order of
    the (row, col) keys and the segment starts) depends only on the dof map: computed on the first call and reused by every
    later assembly on the same space (EMI and each KNP group, and every refresh of a lagged hierarchy)."""
    import threading
    lock = space.__dict__.setdefault("_pattern_lock", threading.Lock())
    with lock:
        pat = getattr(space, "_pattern", None)
        if pat is None and space.dof.size * nd >= 200000:
            # threaded counting sort in the library instead of numpy's argsort of nc nd^2 int64 keys (same order: stable)
            from knpemidg import _abi
            dof32 = np.ascontiguousarray(space.dof, dtype=np.int32)
            nc_, nent = dof32.shape[0], dof32.shape[0] * nd * nd
            order = np.empty(nent, dtype=np.int64)
            starts = np.empty(nent + 1, dtype=np.int64)
            cols = np.empty(nent, dtype=np.int32)
            indptr = np.empty(space.n + 1, dtype=np.int32)
            nseg = np.zeros(1, dtype=np.int64)
            rc = _abi.load().knp_host_block_pattern(nc_, nd, space.n, _abi._p(dof32, _abi._i32p), _abi._p(order, _abi._i64p),
                                                    _abi._p(starts, _abi._i64p), _abi._p(cols, _abi._i32p), _abi._p(indptr, _abi._i32p),
                                                    _abi._p(nseg, _abi._i64p), _setup_threads())
            if rc == 0:
                ns = int(nseg[0])
                pat = space._pattern = (order, starts[:ns].copy(), cols[:ns].copy(), indptr)
        if pat is None:
            dof = space.dof.astype(np.int64)
            key = (np.repeat(dof[:, :, None], nd, axis=2) * space.n + np.repeat(dof[:, None, :], nd, axis=1)).ravel()
            order = np.argsort(key, kind="stable")
            ks = key[order]
            first = np.concatenate([[True], ks[1:] != ks[:-1]])
            starts = np.nonzero(first)[0]
            uk = ks[starts]
            rows = uk // space.n
            indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=space.n))]).astype(np.int32)
            pat = space._pattern = (order, starts, (uk % space.n).astype(np.int32), indptr)
    order, starts, indices, indptr = pat
    src = np.ascontiguousarray(blk, dtype=np.float64).ravel()
    if len(order) >= 200000:
        from knpemidg import _abi
        st = getattr(space, "_pattern_starts64", None)
        if st is None:
            st = space._pattern_starts64 = (np.ascontiguousarray(np.concatenate([starts, [len(order)]]), dtype=np.int64),
                                            np.ascontiguousarray(order, dtype=np.int64))
        data = np.empty(len(starts))
        rc = _abi.load().knp_host_segment_sum(len(starts), _abi._p(st[0], _abi._i64p), _abi._p(st[1], _abi._i64p), _abi._p(src, _abi._f64p),
                                              _abi._p(data, _abi._f64p), _setup_threads())
        if rc != 0:
            data = np.add.reduceat(src[order], starts)
    else:
        data = np.add.reduceat(src[order], starts)
    A = sp.csr_matrix((data, indices, indptr), shape=(space.n, space.n))
    A.has_sorted_indices = True
    return A


# ----------------------------------------------------------------------------------------------
# conforming (membrane-broken) P1 space
# ----------------------------------------------------------------------------------------------
class ConformingSpace:
    def __init__(self, mesh, facet_tags, membrane_tags, nc_owned=None):
        nc = mesh.num_cells()
        fc = mesh.facet_cells
        interior = fc[:, 1] >= 0
        ordinary = interior & ~np.isin(np.asarray(facet_tags), list(membrane_tags))
        g = sp.coo_matrix((np.ones(int(ordinary.sum())), (fc[ordinary, 0], fc[ordinary, 1])), shape=(nc, nc))
        ncomp, comp = csg.connected_components(g, directed=False)
        key = mesh.cells.astype(np.int64) * ncomp + comp[:, None]
        nkeys = int(mesh.coords.shape[0]) * int(ncomp)
        if nkeys <= 8 * key.size:
            # (vertex, component) pairs in use, numbered in ascending key order = what np.unique returns, without its sort of 4 nc keys
            used = np.zeros(nkeys, dtype=bool)
            used[key.ravel()] = True
            ids = np.cumsum(used, dtype=np.int64) - 1
            self.n = int(ids[-1]) + 1 if nkeys else 0
            self.dof = ids[key].astype(np.int32)                   # [nc, nd] conforming dof of each DG dof
        else:
            uniq, inv = np.unique(key.ravel(), return_inverse=True)
            self.n = len(uniq)
            self.dof = inv.reshape(nc, -1).astype(np.int32)
        self.mesh = mesh
        self.ncomp = ncomp

    def cell_blocks(self, coef_nodal, mass_coef=None, cells=None):
        """Per-cell blocks vol * mean(coef) * G (+ mass_coef * vol * M_ref) [n, nv, nv] of all cells, or of `cells`
        (coef / mass_coef are indexed by global cell either way)."""
        d = self.mesh.gdim
        vol, G = _cell_gram(self)
        cbar = coef_nodal.mean(axis=1) if np.ndim(coef_nodal) == 2 else np.asarray(coef_nodal)
        if cells is not None:
            vol, G, cbar = vol[cells], G[cells], cbar[cells]
            mass_coef = None if mass_coef is None else np.asarray(mass_coef)[cells]
        blk = (vol * cbar)[:, None, None] * G
        nv = d + 1
        if mass_coef is not None:
            Mloc = (np.ones((nv, nv)) + np.eye(nv)) / ((d + 1) * (d + 2))
            blk = blk + (mass_coef * vol)[:, None, None] * Mloc[None]
        return blk

    def membrane_blocks(self, membrane):
        """(dofs [F, 2d], blocks [F, 2d, 2d]) of the coupling C int_F (u_i - u_e)(v_i - v_e) on the membrane facets `fids`."""
        mesh = self.mesh
        d = mesh.gdim
        fids, C = membrane
        fcl = mesh.facet_cells[fids]
        fl = mesh.facet_local[fids].astype(np.int64)
        fx = mesh.coords[mesh.facets[fids]]
        if d == 2:
            area = np.linalg.norm(fx[:, 1] - fx[:, 0], axis=1)
        else:
            area = 0.5 * np.linalg.norm(np.cross(fx[:, 1] - fx[:, 0], fx[:, 2] - fx[:, 0]), axis=1)
        # facet vertex m of side s is the cell's local vertex m + (m >= local facet)
        mm = np.arange(d)[None, :]
        d0 = np.take_along_axis(self.dof[fcl[:, 0]], mm + (mm >= fl[:, 0:1]), axis=1)
        d1 = np.take_along_axis(self.dof[fcl[:, 1]], mm + (mm >= fl[:, 1:2]), axis=1)
        Mf = (np.ones((d, d)) + np.eye(d)) / (d * (d + 1))
        blkf = (C * area)[:, None, None] * Mf[None]
        dofs = np.concatenate([d0, d1], axis=1)                              # [F, 2d]
        sgn = np.concatenate([np.ones(d), -np.ones(d)])
        full = np.einsum("a,b,fab->fab", sgn, sgn, np.tile(blkf, (1, 2, 2)))
        return dofs, full

    def stiffness(self, coef_nodal, mass_coef=None, membrane=None):
        """Ac = sum_cells vol * mean(coef) * G (+ mass_coef * M) (+ C int_F jump jump on membrane facets).
        coef_nodal [nc, nd] (kappa) or [nc] (D); membrane = (facet ids, C)."""
        A = _assemble_cached(self, self.cell_blocks(coef_nodal, mass_coef), self.mesh.gdim + 1)
        if membrane is not None:
            dofs, full = self.membrane_blocks(membrane)
            k = dofs.shape[1]
            rows = np.repeat(dofs[:, :, None], k, axis=2).ravel()
            cols = np.repeat(dofs[:, None, :], k, axis=1).ravel()
            A = A + sp.coo_matrix((full.ravel(), (rows, cols)), shape=(self.n, self.n))
        A = A.tocsr()
        A.sum_duplicates()
        return A


# ----------------------------------------------------------------------------------------------
# conforming (membrane-broken) P2 space: auxiliary space of the DG-P2 operators
# ----------------------------------------------------------------------------------------------
class ConformingSpaceP2:
    """Conforming P2 on top of a ConformingSpace (P1).  For DG-P2 the cell-block-Jacobi part over-weights every
    CONTINUOUS quadratic mode by the penalty factor (tau = 20 d p = 120 in 3D), so the auxiliary space must contain
    them: vertex dofs = the P1 space's dofs, edge dofs keyed by the pair of (already membrane-broken) vertex dofs.
    Local order per cell as on the device: vertices, then edges (a, b), a < b, lexicographic.
    The P1 space enters the hierarchy as the first coarse level through `interp` (vertex: identity, edge: mean)."""

    def __init__(self, cs):
        self.cs = cs
        mesh = cs.mesh
        nv = mesh.cells.shape[1]
        self.edges = [(a, b) for a in range(nv) for b in range(a + 1, nv)]
        v = cs.dof.astype(np.int64)
        lo = np.stack([np.minimum(v[:, a], v[:, b]) for a, b in self.edges], axis=1)
        hi = np.stack([np.maximum(v[:, a], v[:, b]) for a, b in self.edges], axis=1)
        uniq, inv = np.unique((lo * cs.n + hi).ravel(), return_inverse=True)
        self.n = cs.n + len(uniq)
        self.dof = np.concatenate([cs.dof, (cs.n + inv.reshape(lo.shape)).astype(np.int32)], axis=1)   # [nc, nd]
        ne = len(uniq)
        rows = np.concatenate([np.arange(cs.n), cs.n + np.arange(ne), cs.n + np.arange(ne)])
        cols = np.concatenate([np.arange(cs.n), uniq // cs.n, uniq % cs.n])
        vals = np.concatenate([np.ones(cs.n), np.full(2 * ne, 0.5)])
        self.interp = sp.csr_matrix((vals, (rows, cols)), shape=(self.n, cs.n))

    def cell_blocks(self, coef, mass_coef=None, cells=None):
        """Per-cell blocks int coef grad u.grad v (+ mass_coef int u v) [n, nd, nd] of all cells, or of `cells`."""
        from knpemidg import dgtab
        from knpemidg.quadrature import simplex_rule
        mesh = self.cs.mesh
        d = mesh.gdim
        vol, Gl = _cell_gram(self.cs)                                         # grad lambda_l . grad lambda_m
        coef = np.asarray(coef, dtype=np.float64)
        if cells is not None:
            vol, Gl, coef = vol[cells], Gl[cells], coef[cells]
            mass_coef = None if mass_coef is None else np.asarray(mass_coef)[cells]
        nd = self.dof.shape[1]
        bary, w = simplex_rule(d, 4)
        B, dB = dgtab.tabulate(2, bary)
        kq = coef @ B.T if coef.ndim == 2 else np.repeat(coef[:, None], len(w), axis=1)      # [nc, q]
        # blk[c, a, b] = vol sum_q w kq dB[q,a,l] Gl[c,l,m] dB[q,b,m]: per point two batched small products
        blk = np.zeros((Gl.shape[0], nd, nd))
        for q in range(len(w)):
            Y = Gl @ dB[q].T                                                  # [nc, nv, nd]
            blk += (w[q] * kq[:, q] * vol)[:, None, None] * (dB[q][None] @ Y)
        if mass_coef is not None:
            Mref = np.einsum("q,qa,qb->ab", w, B, B)
            blk = blk + (np.asarray(mass_coef) * vol)[:, None, None] * Mref[None]
        return blk

    def membrane_blocks(self, membrane):
        """(dofs [F, 2 nd], blocks [F, 2 nd, 2 nd]) of the coupling C int_F jump jump on the membrane facets."""
        from knpemidg import dgtab
        from knpemidg.quadrature import simplex_rule
        mesh = self.cs.mesh
        d = mesh.gdim
        fids, C = membrane
        fcl = mesh.facet_cells[fids]
        fl = mesh.facet_local[fids].astype(np.int64)
        fx = mesh.coords[mesh.facets[fids]]
        if d == 2:
            area = np.linalg.norm(fx[:, 1] - fx[:, 0], axis=1)
        else:
            area = 0.5 * np.linalg.norm(np.cross(fx[:, 1] - fx[:, 0], fx[:, 2] - fx[:, 0]), axis=1)
        mu, wf = simplex_rule(d - 1, 4)
        Bs = np.array([dgtab.tabulate(2, np.insert(mu, i, 0.0, axis=1))[0] for i in range(d + 1)])   # [i, q, nd]
        B0, B1 = Bs[fl[:, 0]], Bs[fl[:, 1]]                                 # [F, q, nd]
        Jm = np.concatenate([B0, -B1], axis=2)                              # jump operator on the facet pair
        full = np.einsum("f,q,fqa,fqb->fab", C * area, wf, Jm, Jm)
        dofs = np.concatenate([self.dof[fcl[:, 0]], self.dof[fcl[:, 1]]], axis=1)
        return dofs, full

    def stiffness(self, coef, mass_coef=None, membrane=None):
        """Conforming-P2 Galerkin operator: sum_cells int coef grad u.grad v (+ mass_coef int u v)
        (+ C int_F jump jump on membrane facets).  coef: [nc, nd] P2 nodal or [nc] cell-wise constant."""
        nd = self.dof.shape[1]
        A = _assemble_cached(self, self.cell_blocks(coef, mass_coef), nd)
        if membrane is not None:
            dofs, full = self.membrane_blocks(membrane)
            rows = np.repeat(dofs[:, :, None], 2 * nd, axis=2).ravel()
            cols = np.repeat(dofs[:, None, :], 2 * nd, axis=1).ravel()
            A = A + sp.coo_matrix((full.ravel(), (rows, cols)), shape=(self.n, self.n))
        A = A.tocsr()
        A.sum_duplicates()
        A.eliminate_zeros()
        return A


# ----------------------------------------------------------------------------------------------
# row-distributed finest level of a partitioned run
# ----------------------------------------------------------------------------------------------
class Dist0Space:
    """Row distribution of the finest conforming level (space: ConformingSpace or ConformingSpaceP2 on the GLOBAL mesh) over the ranks of
    a cell partition `owner` [nc].  The reference's BoomerAMG is row-distributed by PETSc (src/knpemidg/solver.py:433, 688); here only the
    finest level is, in the sub-assembled form of non-overlapping domain decomposition (csrc/amg.hip: dist0):

    * rank r's ELEMENTS are its owned cells and the membrane facets whose first cell it owns; its level-0 rows `verts` are the conforming
      dofs those elements touch (ascending global number = local number);
    * its level-0 matrix is assembled from its elements only, so sum_r S_r^T A_r S_r = A (S_r: selection of the rank's rows) and a product
      with an accumulated vector is a per-rank partial sum;
    * dofs touched by elements of several ranks are SHARED: `interface_tables` lists them per peer in ascending global order (both sides
      agree without communication: every rank computes all of this from the global mesh).
    """

    def __init__(self, space, owner, membrane_facets, rank, world):
        self.space = space
        self.rank, self.world = int(rank), int(world)
        self.owner = np.asarray(owner)
        self.mesh = (space.cs if hasattr(space, "cs") else space).mesh
        self.mem = np.asarray(membrane_facets, dtype=np.int64)
        self.facet_owner = self.owner[self.mesh.facet_cells[self.mem, 0]] if len(self.mem) else np.zeros(0, dtype=self.owner.dtype)
        mdofs = space.membrane_blocks((self.mem, 1.0))[0] if len(self.mem) else None
        self.rank_verts = []
        for q in range(self.world):
            parts = [space.dof[self.owner == q].ravel()]
            if mdofs is not None:
                parts.append(mdofs[self.facet_owner == q].ravel())
            self.rank_verts.append(np.unique(np.concatenate(parts)).astype(np.int64))
        self.verts = self.rank_verts[self.rank]
        self.n = len(self.verts)
        self.g2l = np.full(space.n, -1, dtype=np.int64)
        self.g2l[self.verts] = np.arange(self.n)
        self.cells = np.nonzero(self.owner == self.rank)[0]
        self.facets = self.mem[self.facet_owner == self.rank] if len(self.mem) else self.mem

    def local_dg2cg(self, cells_global):
        """[n_local_cells, nd] local conforming dof of every DG dof of the rank's cells (owned first, then ghosts); dofs of ghost cells
        outside the rank's rows map to 0 (never read: vector updates and the prolongation touch owned cells only)."""
        d = self.g2l[self.space.dof[cells_global]]
        return np.where(d >= 0, d, 0).astype(np.int32)

    def local_matrix(self, coef, mass_coef=None, membrane_C=None):
        """The rank's sub-assembled level-0 matrix [n, n] (same arguments as space.stiffness; membrane_C: the coupling constant)."""
        sp_ = self.space
        blk = sp_.cell_blocks(coef, mass_coef, cells=self.cells)
        dofs = self.g2l[sp_.dof[self.cells]]
        k = dofs.shape[1]
        rows = [np.repeat(dofs[:, :, None], k, axis=2).ravel()]
        cols = [np.repeat(dofs[:, None, :], k, axis=1).ravel()]
        vals = [blk.ravel()]
        if membrane_C is not None and len(self.facets):
            fd, full = sp_.membrane_blocks((self.facets, float(membrane_C)))
            fd = self.g2l[fd]
            k = fd.shape[1]
            rows.append(np.repeat(fd[:, :, None], k, axis=2).ravel())
            cols.append(np.repeat(fd[:, None, :], k, axis=1).ravel())
            vals.append(full.ravel())
        A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.n, self.n)).tocsr()
        A.sum_duplicates()
        A.sort_indices()
        return A

    def localize(self, levels, A_local):
        """The hierarchy with level 0 replaced by the rank's rows: A sub-assembled, inverse diagonal / prolongator rows of the GLOBAL level
        (so the smoother and the Galerkin coarse operators are exactly those of the replicated hierarchy).  None if there is no coarser level."""
        if len(levels) < 2:
            return None
        g = levels[0]
        lv = Level()
        lv.A = A_local
        lv.dinv = np.ascontiguousarray(g.dinv[self.verts])
        lv.rho, lv.cheb_degree, lv.cheb_lower = g.rho, g.cheb_degree, g.cheb_lower
        lv.P = g.P.tocsr()[self.verts].tocsr()
        lv.P.sort_indices()
        lv.R = lv.P.T.tocsr()
        lv.R.sort_indices()
        return [lv] + list(levels[1:])

    def interface_tables(self):
        """(peers, lists, uvtx, aptr, asrc) for Device.amg_interface: lists[p] = local numbers of the dofs shared with peers[p] in ascending
        global order; uvtx = the distinct shared dofs (local); per distinct dof aptr / asrc give the message positions of the other owners'
        values in ascending rank order, -1 standing for this rank's own value."""
        peers, lists, g_all, r_all = [], [], [], []
        for q in range(self.world):
            if q == self.rank:
                continue
            sh = np.intersect1d(self.verts, self.rank_verts[q], assume_unique=True)
            if len(sh):
                peers.append(q)
                lists.append(self.g2l[sh].astype(np.int32))
                g_all.append(sh)
                r_all.append(np.full(len(sh), q, dtype=np.int64))
        if not peers:
            return [], [], np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32)
        g = np.concatenate(g_all)
        rk = np.concatenate(r_all)
        pos = np.arange(len(g), dtype=np.int64)
        ug = np.unique(g)
        g2 = np.concatenate([g, ug])
        rk2 = np.concatenate([rk, np.full(len(ug), self.rank, dtype=np.int64)])
        pos2 = np.concatenate([pos, np.full(len(ug), -1, dtype=np.int64)])
        order = np.lexsort((rk2, g2))
        cnt = np.bincount(np.searchsorted(ug, g2), minlength=len(ug))
        aptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        return peers, lists, self.g2l[ug].astype(np.int32), aptr, pos2[order].astype(np.int32)


# ----------------------------------------------------------------------------------------------
# smoothed aggregation
# ----------------------------------------------------------------------------------------------
def _rowmax(S, vals):
    """max over the stored neighbours of each row of vals[col]; -inf for empty rows."""
    out = np.full(S.shape[0], -np.inf)
    nz = np.diff(S.indptr) > 0
    if S.nnz:
        red = np.maximum.reduceat(vals[S.indices], S.indptr[:-1][nz])
        out[nz] = red
    return out


def mis2_aggregate(S, seed=0):
    """Aggregation from a distance-2 maximal independent set of the strength graph S (symmetric pattern,
    no diagonal).  Vectorised Luby rounds (the same algorithm maps 1:1 onto GPU kernels).
    Returns agg[n] in [0, nagg)."""
    n = S.shape[0]
    rng = np.random.default_rng(seed)
    key = rng.permutation(n).astype(np.float64) + 1.0
    import os
    if n >= 20000 and os.environ.get("KNP_SETUP_NATIVE_MIS2", "1") != "0":
        # the same rounds in the library (csrc/host_sparse.cpp: knp_host_mis2_aggregate), without the gather + reduceat temporaries
        from knpemidg import _abi
        ip, ix = np.ascontiguousarray(S.indptr, dtype=np.int32), np.ascontiguousarray(S.indices, dtype=np.int32)
        agg = np.empty(n, dtype=np.int64)
        nagg = int(_abi.load().knp_host_mis2_aggregate(n, _abi._p(ip, _abi._i32p), _abi._p(ix, _abi._i32p), _abi._p(key, _abi._f64p),
                                                        _abi._p(agg, _abi._i64p), _setup_threads()))
        if nagg >= 0:
            return agg, nagg
    state = np.zeros(n, dtype=np.int8)                 # 0 undecided, 1 root, 2 covered
    isolated = np.diff(S.indptr) == 0
    state[isolated] = 1
    while (state == 0).any():
        k = np.where(state == 0, key, -np.inf)
        m1 = np.maximum(k, _rowmax(S, k))
        m2 = np.maximum(m1, _rowmax(S, m1))
        new_root = (state == 0) & (k >= m2)
        state[new_root] = 1
        r = np.where(state == 1, 1.0, -np.inf)
        c1 = np.maximum(r, _rowmax(S, r))
        c2 = np.maximum(c1, _rowmax(S, c1))
        state[(state == 0) & (c2 > 0)] = 2
    roots = np.nonzero(state == 1)[0]
    agg = np.full(n, -1, dtype=np.int64)
    agg[roots] = np.arange(len(roots))
    # distance-1 members take the neighbouring root with the largest key, distance-2 members follow a neighbour
    for _ in range(2):
        lab = np.where(agg >= 0, key * 0 + agg.astype(np.float64), -np.inf)
        best = _rowmax(S, lab)
        take = (agg < 0) & np.isfinite(best)
        agg[take] = best[take].astype(np.int64)
    left = agg < 0
    if left.any():                                     # safety: singletons
        agg[left] = len(roots) + np.arange(int(left.sum()))
    return agg, int(agg.max()) + 1


def _threaded_matvec(A):
    """x -> A x through the library's threaded CSR product (csrc/host_sparse.cpp) for the large levels, scipy's for the small ones."""
    if A.shape[0] < 50000:
        return lambda x: A @ x
    from knpemidg import _abi
    A = A.tocsr()
    ip, ix, dv = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data, dtype=np.float64)
    lib = _abi.load()
    nthreads = _setup_threads()

    def mv(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(A.shape[0])
        lib.knp_host_spmv(A.shape[0], _abi._p(ip, _abi._i32p), _abi._p(ix, _abi._i32p), _abi._p(dv, _abi._f64p), _abi._p(x, _abi._f64p),
                          _abi._p(y, _abi._f64p), nthreads)
        return y
    return mv


def _setup_threads():
    """Threads of the host setup kernels: the cores this process may use, at most 16 (KNP_SETUP_THREADS overrides)."""
    import os
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return int(os.environ.get("KNP_SETUP_THREADS", max(1, min(16, avail))))


def _spectral_radius_DinvA(A, dinv, iters=15, seed=1):
    x = np.random.default_rng(seed).standard_normal(A.shape[0])
    lam = 1.0
    mv = _threaded_matvec(A)
    for _ in range(iters):
        y = dinv * mv(x)
        lam = np.linalg.norm(y) / max(np.linalg.norm(x), 1e-300)
        x = y / max(np.linalg.norm(y), 1e-300)
    return 1.1 * lam


class Level:
    pass


def _spgemm(A, B):
    """Sparse product through the library's threaded Gustavson kernel (csrc/host_sparse.cpp)."""
    from knpemidg import _abi
    return _abi.host_spgemm(A, B)


def _native_sa(n):
    """The library's smoothed-aggregation passes for levels of at least 20 000 rows (KNP_SETUP_NATIVE_SA=0: the numpy / scipy lines)."""
    import os
    return n >= 20000 and os.environ.get("KNP_SETUP_NATIVE_SA", "1") != "0"


def _csr_out(lib, n, m, Cp, cj, cx=None):
    """scipy CSR from row pointers filled by the library and its malloc'ed column / value arrays (copied, then released)."""
    from knpemidg import _abi
    nnz = int(Cp[-1])
    try:
        Cj = np.ctypeslib.as_array(cj, shape=(max(nnz, 1),))[:nnz].copy()
        Cx = np.ones(nnz) if cx is None else np.ctypeslib.as_array(cx, shape=(max(nnz, 1),))[:nnz].copy()
    finally:
        lib.knp_host_free(cj)
        if cx is not None:
            lib.knp_host_free(cx)
    out = sp.csr_matrix((Cx, Cj, Cp), shape=(n, m))
    out.has_sorted_indices = True
    return out


def _csr_in(A):
    from knpemidg import _abi
    A = A.tocsr()
    return (np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32),
            np.ascontiguousarray(A.data, dtype=np.float64))


def _strength_native(A, d, theta):
    from knpemidg import _abi
    lib = _abi.load()
    n = A.shape[0]
    Ap, Aj, Ax = _csr_in(A)
    dd = np.ascontiguousarray(d, dtype=np.float64)
    Sp, sj = np.empty(n + 1, dtype=np.int32), _abi._i32p()
    if lib.knp_host_strength(n, _abi._p(Ap, _abi._i32p), _abi._p(Aj, _abi._i32p), _abi._p(Ax, _abi._f64p), _abi._p(dd, _abi._f64p), float(theta),
                             _abi._p(Sp, _abi._i32p), _abi.C.byref(sj), _setup_threads()) != 0:
        return None
    return _csr_out(lib, n, n, Sp, sj)


def _smooth_native(A, v, P):
    from knpemidg import _abi
    lib = _abi.load()
    n, m = P.shape
    Ap, Aj, Ax = _csr_in(A)
    Pp, Pj, Px = _csr_in(P)
    vv = np.ascontiguousarray(v, dtype=np.float64)
    Cp, cj, cx = np.empty(n + 1, dtype=np.int32), _abi._i32p(), _abi._f64p()
    if lib.knp_host_smooth_prolongator(n, m, _abi._p(Ap, _abi._i32p), _abi._p(Aj, _abi._i32p), _abi._p(Ax, _abi._f64p), _abi._p(vv, _abi._f64p),
                                       _abi._p(Pp, _abi._i32p), _abi._p(Pj, _abi._i32p), _abi._p(Px, _abi._f64p), _abi._p(Cp, _abi._i32p),
                                       _abi.C.byref(cj), _abi.C.byref(cx), _setup_threads()) != 0:
        return None
    return _csr_out(lib, n, m, Cp, cj, cx)


def _truncate_native(P, trunc, Bc):
    from knpemidg import _abi
    lib = _abi.load()
    n, m = P.shape
    Pp, Pj, Px = _csr_in(P)
    bc = np.ascontiguousarray(Bc, dtype=np.float64)
    Tp, tj, tx = np.empty(n + 1, dtype=np.int32), _abi._i32p(), _abi._f64p()
    if lib.knp_host_truncate_prolongator(n, _abi._p(Pp, _abi._i32p), _abi._p(Pj, _abi._i32p), _abi._p(Px, _abi._f64p), float(trunc),
                                         _abi._p(bc, _abi._f64p), _abi._p(Tp, _abi._i32p), _abi.C.byref(tj), _abi.C.byref(tx), _setup_threads()) != 0:
        return None
    return _csr_out(lib, n, m, Tp, tj, tx)


def _scale_rows(A, v):
    A = A.tocsr()
    return sp.csr_matrix((A.data * np.repeat(v, np.diff(A.indptr)), A.indices, A.indptr), shape=A.shape)


def build_hierarchy(A, theta=0.08, max_coarse=4000, max_levels=12, cheb_degree=None, cheb_lower=0.3, psmooth=2, trunc=0.04,
                    top_interp=None, top_degree=2, top_lower=0.1, level0_degree=None):
    """Smoothed-aggregation hierarchy for an SPD (possibly singular, constants) matrix.
    Each level: A (csr), dinv, rho = spectral radius estimate of D^-1 A, P (csr, to the next level).
    Last level: dense pseudo-inverse.  top_interp: geometric prolongator of the first level (conforming P1 -> P2,
    ConformingSpaceP2.interp); aggregation starts below it."""
    import os
    theta = float(os.environ.get("KNP_AMG_THETA", theta))
    # smoother degree of the aggregated levels: 1 since round 3 (with the cheaper applies and the fused restriction the V-cycle is a third
    # of a step; degree 1 instead of 2 takes two SpMV kernels out of every level visit and costs no iterations: r=2 8.23 -> 7.85 ms/step,
    # EMI 6.55 -> 6.35 / KNP 5.7 -> 5.75 iterations, degree 3 8.41; profiles/r03_amg_degree_sweep.txt).  The DG-P2 hierarchies (top_interp)
    # keep degree 2: with degree 1 the P2 configuration steps 4 % faster, but its EMI solve meets the stopping test on the preconditioned
    # norm after two iterations with 1.35e-6 left in the concentrations (test_production_tolerances_r1_against_tight_solves[2])
    # (round 4: degree 1 for the DG-P2 hierarchies too -- the EMI stop no longer depends on the preconditioner, profiles/r04_stop_sweep.txt)
    if cheb_degree is None:
        cheb_degree = 1
    cheb_degree = int(os.environ.get("KNP_AMG_DEGREE", cheb_degree))
    cheb_lower = float(os.environ.get("KNP_AMG_LOWER", cheb_lower))
    max_coarse = int(os.environ.get("KNP_AMG_MAXCOARSE", max_coarse))
    trunc = float(os.environ.get("KNP_AMG_TRUNC", trunc))
    levels = []
    A = A.tocsr().astype(np.float64)
    Bnull = np.ones(A.shape[0])
    while True:
        lv = Level()
        lv.A = A
        d = A.diagonal()
        lv.dinv = 1.0 / d
        lv.rho = _spectral_radius_DinvA(A, lv.dinv)
        lv.cheb_degree, lv.cheb_lower = cheb_degree, cheb_lower
        if levels and "KNP_AMG_COARSE_DEGREE" in os.environ:          # experiment knob: smoother degree below level 0
            lv.cheb_degree = int(os.environ["KNP_AMG_COARSE_DEGREE"])
        if not levels and top_interp is None and level0_degree is not None:
            # finest conforming-P1 level: the DG block-Jacobi part of the preconditioner already damps what a smoother on
            # this level would; 0 = transfer-only level (x = P x_coarse), 1 = one damped-Jacobi step before / after
            lv.cheb_degree = int(os.environ.get("KNP_AMG_DEGREE0", level0_degree))
        levels.append(lv)
        n = A.shape[0]
        if top_interp is not None and len(levels) == 1:
            # the conforming-P2 level is smoothed on [0.1 rho, rho] (P2 stiffness + mass matrices have a wider Jacobi-scaled
            # spectrum than P1).  Degree 3 was the best with the assembled applies of round 1; with the matrix-free applies the
            # V-cycle is 55 % of a P2 step and degree 2 wins: KNP 10.2 -> 11.7 iterations, 11.7 -> 11.3 ms/step at r=1
            lv.cheb_degree = int(os.environ.get("KNP_AMG_TOPDEGREE", top_degree or cheb_degree))
            lv.cheb_lower = float(os.environ.get("KNP_AMG_TOPLOWER", top_lower or cheb_lower))
            P = top_interp.tocsr().astype(np.float64)
            P.sort_indices()
            lv.P = P
            lv.R = P.T.tocsr()
            lv.R.sort_indices()
            A = _spgemm(lv.R, _spgemm(A, P))
            Bnull = np.ones(A.shape[0])          # the geometric interpolation reproduces constants exactly
            continue
        if n <= max_coarse or len(levels) >= max_levels:
            break
        # symmetric strength of connection
        A.sort_indices()
        native = _native_sa(n)
        S = _strength_native(A, d, theta) if native else None
        if S is None:
            rowid = np.repeat(np.arange(n), np.diff(A.indptr))
            strong = (rowid != A.indices) & (np.abs(A.data) >= theta * np.sqrt(np.abs(d[rowid] * d[A.indices])))
            S = sp.csr_matrix((np.ones(int(strong.sum())), A.indices[strong],
                               np.concatenate([[0], np.cumsum(np.bincount(rowid[strong], minlength=n))])), shape=A.shape)
        agg, nagg = mis2_aggregate(S, seed=len(levels))
        if nagg >= n:
            break
        # tentative prolongator from the near-null-space candidate Bnull of THIS level (the constant on level 0; on
        # coarser levels its coarse representation, which is not constant because the aggregates differ in size):
        # T[i, agg(i)] = Bnull[i] / |Bnull restricted to the aggregate|, and the next level's candidate is that norm, so
        # that T Bnull_coarse = Bnull on every level
        nrm = np.sqrt(np.bincount(agg, weights=Bnull * Bnull, minlength=nagg))
        T = sp.csr_matrix((Bnull / nrm[agg], (np.arange(n), agg)), shape=(n, nagg))
        # prolongator smoothing: `psmooth` damped-Jacobi steps on the tentative prolongator.  Two steps instead of the
        # textbook one cut the V-cycle's convergence factor on the anisotropic conforming operator from ~0.63 to
        # ~0.25 (CG on Ac: 39 -> 13 iterations at r=1) for 1.8x the operator complexity
        omega = (4.0 / 3.0) / lv.rho
        P = T
        DA = None
        for _ in range(psmooth):
            Pn = _smooth_native(A, omega * lv.dinv, P) if native else None      # P - (omega D^-1 A) P in one pass of the library
            if Pn is None:
                if DA is None:
                    DA = _scale_rows(A, omega * lv.dinv)                   # omega D^-1 A, once per level
                Pn = (P - _spgemm(DA, P)).tocsr()
            P = Pn
        Pt = _truncate_native(P, trunc, nrm) if (trunc > 0 and native) else None
        if Pt is not None:
            P = Pt
        elif trunc > 0:
            # prolongator truncation: drop entries below trunc * (row maximum) and rescale every row so that the
            # coarse near-null-space vector is still interpolated to the same fine values; keeps the convergence of
            # the smoothed prolongator at less than half its operator complexity (3.84 -> 1.57 at r=1)
            P.sort_indices()
            absd = np.abs(P.data)
            nzr = np.diff(P.indptr) > 0
            rowmax = np.zeros(n)
            rowmax[nzr] = np.maximum.reduceat(absd, P.indptr[:-1][nzr])
            rowid = np.repeat(np.arange(n), np.diff(P.indptr))
            keep = absd >= trunc * rowmax[rowid]
            Pt = sp.csr_matrix((P.data[keep], P.indices[keep], np.concatenate([[0], np.cumsum(np.bincount(rowid[keep], minlength=n))])),
                               shape=P.shape)
            Bc = nrm
            tgt, got = P @ Bc, Pt @ Bc
            P = (sp.diags(np.where(np.abs(got) > 1e-300, tgt / np.where(got == 0, 1.0, got), 1.0)) @ Pt).tocsr()
        P.sort_indices()
        lv.P = P
        lv.R = P.T.tocsr()
        lv.R.sort_indices()
        A = _spgemm(lv.R, _spgemm(A, P))
        Bnull = nrm
    last = levels[-1]
    last.pinv = _coarse_pseudo_inverse(last.A, Bnull)
    return levels


def _coarse_pseudo_inverse(A, Bnull):
    """Dense (pseudo-)inverse of the coarsest operator, rounded to fp32 (the device stores it in fp32).
    The constant mode of the singular EMI operator reaches the coarsest level as a tiny but non-zero eigenvalue (smoothed
    prolongators do not reproduce constants to rounding): inverting it would put a huge component into the V-cycle.  With the
    coarse near-null-space candidate n (carried down the hierarchy, |n| = 1) the deflated inverse
        A^+  =  (A + s n n^T)^-1  -  n n^T / s ,      s = mean diagonal,
    is exact for A n = 0 and costs one Cholesky-based inverse (LAPACK potrf / potri, threaded) instead of a full
    eigen-decomposition -- 19 s -> 0.6 s for the 3 089-dof coarsest level of the r=2 mesh.  The nonsingular KNP operators
    (mass term) take the plain inverse.  Anything else (several null vectors: subdomains without any coupling) falls back
    to the eigen-decomposition, where everything below 1e-9 of the largest eigenvalue counts as null space."""
    import scipy.linalg as sla
    A = A.tocsr()
    N = A.shape[0]
    nvec = np.asarray(Bnull, dtype=np.float64) / max(np.linalg.norm(Bnull), 1e-300)
    s = float(A.diagonal().sum()) / max(N, 1)
    rayleigh = float(nvec @ (A @ nvec))
    singular = rayleigh < 1e-7 * s                                 # (near-)singular along the candidate
    Pi = None
    # Every dense pass over the N x N matrix (76 MB at N = 3 089) costs as much as a tenth of the factorisation: the matrix is built in
    # Fortran order (no LAPACK copy), only its lower triangle is touched (rank-one updates by BLAS dsyr, potrf / potri in place) and the
    # mirror + fp32 rounding + sanity checks are one threaded pass in the library.
    try:
        M = A.toarray(order="F")
        if singular:
            M = sla.blas.dsyr(s, nvec, a=M, lower=1, overwrite_a=1)
        M, info = sla.lapack.dpotrf(M, lower=1, overwrite_a=1, clean=0)
        if info != 0:
            raise np.linalg.LinAlgError("potrf")
        M, info = sla.lapack.dpotri(M, lower=1, overwrite_c=1)     # no identity right-hand side: half the flops of a solve with it
        if info != 0:
            raise np.linalg.LinAlgError("potri")
        if singular:
            M = sla.blas.dsyr(-1.0 / s, nvec, a=M, lower=1, overwrite_a=1)
        Pi = _mirror_round(M)
        if Pi is not None and not (Pi[1] * s <= 1e10):             # NaN / Inf / another (near-)null vector hiding in there
            Pi = None
        M = None
    except (np.linalg.LinAlgError, sla.LinAlgError, ValueError):
        Pi = None
    if Pi is not None:
        return Pi[0]
    Ad = A.toarray()
    Ad = 0.5 * (Ad + Ad.T)
    w, V = np.linalg.eigh(Ad)
    keep = w > 1e-9 * w.max()
    Pi = (V[:, keep] / w[keep]) @ V[:, keep].T
    # symmetrise before rounding so that the fp32 operator is still exactly symmetric
    return (0.5 * (Pi + Pi.T)).astype(np.float32)


def _mirror_round(M):
    """(full symmetric fp32 matrix, max |entry|) from a Fortran-order matrix whose LOWER triangle is valid (LAPACK potri output)."""
    N = M.shape[0]
    Mc = M.T                                                       # C-contiguous view: the valid triangle is its upper one
    if Mc.flags.c_contiguous and N >= 256:
        from knpemidg import _abi
        try:
            out = np.empty((N, N), dtype=np.float32)
            mx = np.zeros(1)
            if _abi.load().knp_host_sym_to_f32(N, _abi._p(Mc, _abi._f64p), _abi._p(out, _abi._f32p), _abi._p(mx, _abi._f64p), _setup_threads()) == 0:
                return out, float(mx[0])
        except (OSError, AttributeError):
            pass
    L = np.tril(np.asarray(M))
    full = L + L.T
    full[np.diag_indices(N)] *= 0.5
    mx = float(np.abs(full).max()) if N else 0.0
    return full.astype(np.float32), (mx if np.isfinite(full).all() else float("nan"))


def build_emi_levels(cspace, cspace2, facet_tags, membrane_tags, kappa, C_phi):
    """Host part of the EMI preconditioner setup (the reference builds BoomerAMG from BB_emi, solver.py:433, 505): conforming
    operator from kappa ([nc, nd] nodal or [nc]) + the membrane coupling C_phi, and its smoothed-aggregation hierarchy.
    Pure host code: runs in the solver's process (refreshes, distributed setup) or in its helper process (first build)."""
    import os
    mesh = cspace.mesh
    mem = np.nonzero((mesh.facet_cells[:, 1] >= 0) & np.isin(np.asarray(facet_tags), list(membrane_tags)))[0]
    psmooth = int(os.environ.get("KNP_AMG_PSMOOTH_EMI", 3))
    if cspace2 is None:
        Ac = cspace.stiffness(kappa, membrane=(mem, float(C_phi)))
        # EMI: the weakly coupled, long and thin intracellular tubes need wide interpolation: three damped-Jacobi steps on the
        # tentative prolongator (PCG iterations at r=2: 68 / 17 / 8 for 1 / 2 / 3 steps; on the conforming problem alone two
        # steps lose mesh independence, 12 -> 24 from r=1 to r=2, three do not); no smoother on the finest conforming level
        # (same iteration count with or without it: block-Jacobi on the DG space does that job)
        # finest conforming level: ONE damped-Jacobi step before / after (round 3; rounds 1-2: transfer-only).  With the round-3 kernel costs
        # the extra level-0 kernels are cheap against what they save: r=2 EMI 6.35 -> 4.25 PCG iterations (7.85 -> 7.43 ms/step), r=3 8.6 ->
        # 6.8, and on the unstructured EMIx reconstruction 32.3 -> 11.7 (8.04 -> 5.94 ms/step); two steps buy nothing more
        # (profiles/r03_amg_degree_sweep.txt).  A partitioned run then all-reduces the level-0 restricted residual (as KNP does).
        return build_hierarchy(Ac, psmooth=psmooth, level0_degree=int(os.environ.get("KNP_AMG_DEGREE0_EMI", 1)))
    # DG-P2: auxiliary space = conforming P2 (block-Jacobi over-weights continuous quadratics by the penalty factor); the
    # conforming P1 space is its first coarse level, aggregation starts below
    Ac = cspace2.stiffness(kappa, membrane=(mem, float(C_phi)))
    return build_hierarchy(Ac, psmooth=psmooth, top_interp=cspace2.interp)


def build_knp_groups(cspace, cspace2, sub_tags, D_subs, dt, level0_degree):
    """Host part of the KNP preconditioner setup (reference: BoomerAMG on AA_knp, solver.py:688, 767): per species (or per group
    of species with close diffusion coefficients) the conforming operator  1/dt M + D_k K  (symmetric part; the drift enters only
    the Krylov operator) and its smoothed-aggregation hierarchy.  cspace2: the conforming P2 space for degree 2, else None.
    D_subs: one {cell tag: D} dict per solved species.  Returns [(members, levels)].  Pure host code: runs in the solver's
    process or in its helper process (knpemidg/setup_worker.py)."""
    import os
    tags = np.asarray(sub_tags)
    nc = len(tags)
    Dk = []
    for d in D_subs:
        q = np.zeros(nc, dtype=np.float64)
        for key, value in d.items():
            q[tags == int(key)] = float(value)
        Dk.append(q)
    # species whose diffusion coefficients differ by < 25 % share one hierarchy built from the mean coefficient (it only
    # preconditions): one chain of V-cycle kernels then carries all species as right-hand-side columns instead of one
    # concurrent chain per species
    pos = Dk[0] > 0
    shared = (len(Dk) > 1 and os.environ.get("KNP_AMG_SHARED", "1") == "1"
              and all(np.all(np.abs(D[pos] / Dk[0][pos] - 1.0) < 0.25) and np.all((D > 0) == pos) for D in Dk[1:]))
    groups = [(list(range(len(Dk))), np.mean(Dk, axis=0))] if shared else [([k], D) for k, D in enumerate(Dk)]
    psmooth = int(os.environ.get("KNP_AMG_PSMOOTH_KNP", 2))
    out = []
    for members, D in groups:
        mass = np.full(nc, 1.0 / float(dt))
        if cspace2 is None:
            Ac = cspace.stiffness(D, mass_coef=mass)
            levels = build_hierarchy(Ac, psmooth=psmooth, level0_degree=int(os.environ.get("KNP_AMG_DEGREE0_KNP", level0_degree)))
        else:
            Ac = cspace2.stiffness(D, mass_coef=mass)
            levels = build_hierarchy(Ac, psmooth=psmooth, top_interp=cspace2.interp)
        out.append((members, levels))
    return out


def cheb_coefficients(rho, degree, lower):
    """Chebyshev polynomial smoother on D^-1 A for eigenvalues in [lower*rho, rho] (Saad, Alg. 12.1)."""
    lmax, lmin = rho, lower * rho
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    return theta, delta
