"""Solution-independent right-hand-side terms of the manufactured-solution problem, integrated on the host for
DG-p, p = 1, 2 (reference: src/knpemidg/solver.py:349-374 for L_emi, 632-657 for L_knp).  The `mms` object and the ion
dictionaries use the reference's keys; every entry that is a UFL expression there is a callable f(X[..., d]) here."""
import numpy as np

from knpemidg.quadrature import simplex_rule
from knpemidg.dgtab import tabulate

QDEG = 8


def _geometry(mesh):
    d = mesh.gdim
    x = mesh.coords[mesh.cells]
    J = (x[:, 1:, :] - x[:, :1, :]).transpose(0, 2, 1)
    vol = np.abs(np.linalg.det(J)) / {2: 2.0, 3: 6.0}[d]
    fx = mesh.coords[mesh.facets]
    if d == 2:
        t = fx[:, 1] - fx[:, 0]
        area = np.linalg.norm(t, axis=1)
        n = np.stack([t[:, 1], -t[:, 0]], axis=1)
    else:
        n = np.cross(fx[:, 1] - fx[:, 0], fx[:, 2] - fx[:, 0])
        area = 0.5 * np.linalg.norm(n, axis=1)
    n = n / np.linalg.norm(n, axis=1)[:, None]
    c0 = mesh.facet_cells[:, 0]
    l0 = mesh.facet_local[:, 0].astype(np.int64)
    apex = mesh.coords[mesh.cells[c0, l0]]
    n *= np.sign(np.einsum("fd,fd->f", n, fx[:, 0] - apex))[:, None]        # outward from side 0
    return vol, area, n


def _ev(f, X):
    return np.broadcast_to(np.asarray(f(X), dtype=np.float64), X.shape[:-1])


def _cell_source(mesh, vol, sel, f, out, p=1):
    """out[c, a] += int_c f phi_a for the selected cells."""
    d = mesh.gdim
    bary, w = simplex_rule(d, QDEG)
    X = np.einsum("ql,cld->cqd", bary, mesh.coords[mesh.cells[sel]])
    out[sel] += np.einsum("q,c,cq,qa->ca", w, vol[sel], _ev(f, X), tabulate(p, bary)[0])


def _facet_term(mesh, area, fids, side, g, out, sign=1.0, p=1):
    """out[cell(side), a] += sign * int_F g phi_a  for facets fids (side: array of 0/1 per facet)."""
    if len(fids) == 0:
        return
    d = mesh.gdim
    mu, w = simplex_rule(d - 1, QDEG)
    X = np.einsum("ql,fld->fqd", mu, mesh.coords[mesh.facets[fids]])
    cells = mesh.facet_cells[fids, side]
    lf = mesh.facet_local[fids, side].astype(np.int64)
    # trace of the cell basis on local facet i: facet vertex m is the cell's local vertex m + (m >= i)
    Bs = np.array([tabulate(p, np.insert(mu, i, 0.0, axis=1))[0] for i in range(d + 1)])       # [i, q, nd]
    vals = np.einsum("q,f,fq,fqa->fa", w, area[fids], _ev(g, X), Bs[lf]) * sign
    np.add.at(out, cells, vals)


def extra_rhs(solver):
    """(C[n_sys, nc], extra_emi[nc, nd], extra_knp[n_sys, nc, nd])."""
    mesh, mms = solver.mesh, solver.mms
    tags = solver.subdomains.array()
    ft = solver.surfaces.array()
    nc, nd = mesh.num_cells(), solver.nd
    p = solver.degree_knp
    vol, area, normal = _geometry(mesh)
    F, C_phi = float(solver.F), float(solver.C_phi)
    plus_side = solver.n_g.plus_side.astype(np.int64)
    ics, ecs = np.nonzero(tags == 1)[0], np.nonzero(tags == 0)[0]
    ext = np.nonzero(mesh.facet_cells[:, 1] < 0)[0]

    e_emi = np.zeros((nc, nd))
    _cell_source(mesh, vol, ics, mms.rhs['volume_phi_1'], e_emi, p=p)                    # solver.py:365
    _cell_source(mesh, vol, ecs, mms.rhs['volume_phi_2'], e_emi, p=p)                    # solver.py:366
    for tag in solver.lm_tags:
        fids = np.nonzero((ft == tag) & (mesh.facet_cells[:, 1] >= 0))[0]
        ps = plus_side[fids]
        g = mms.rhs['bdry']['u_phi'][tag]
        _facet_term(mesh, area, fids, 1 - ps, lambda X: C_phi * _ev(g, X), e_emi, +1.0, p=p)      # C_phi g minus(v)
        _facet_term(mesh, area, fids, ps, lambda X: C_phi * _ev(g, X), e_emi, -1.0, p=p)          # - C_phi g plus(v)   (359)
        _facet_term(mesh, area, fids, ps, mms.rhs['bdry']['stress'][tag], e_emi, +1.0, p=p)       # g_stress plus(v)    (369)
    zero_side = np.zeros(len(ext), dtype=np.int64)
    for ion in solver.ion_list:                                                               # solver.py:372-374
        z = float(ion['z'])
        _facet_term(mesh, area, ext, zero_side,
                    lambda X, ion=ion: np.einsum("fqd,fd->fq", np.asarray(ion['bdry'](X)), normal[ext]), e_emi, -F * z, p=p)

    ns = solver.N_ions
    Cdev = np.zeros((ns, nc))
    e_knp = np.zeros((ns, nc, nd))
    for k, ion in enumerate(solver.ion_list[:-1]):
        Cdev[k] = ion['C']
        _cell_source(mesh, vol, ics, ion['f1'], e_knp[k], p=p)                            # solver.py:645
        _cell_source(mesh, vol, ecs, ion['f2'], e_knp[k], p=p)                            # solver.py:646
        C1, C2 = float(ion['C_sub'][1]), float(ion['C_sub'][0])
        for tag in solver.lm_tags:
            fids = np.nonzero((ft == tag) & (mesh.facet_cells[:, 1] >= 0))[0]
            ps = plus_side[fids]
            _facet_term(mesh, area, fids, 1 - ps, ion['g_robin_1'][tag], e_knp[k], +C1, p=p)       # C_1 g_1 minus(v)   (653)
            _facet_term(mesh, area, fids, ps, ion['g_robin_2'][tag], e_knp[k], -C2, p=p)           # - C_2 g_2 plus(v)  (654)
        _facet_term(mesh, area, ext, zero_side,
                    lambda X, ion=ion: np.einsum("fqd,fd->fq", np.asarray(ion['bdry'](X)), normal[ext]), e_knp[k], -1.0, p=p)   # (657)
    return Cdev, e_emi, e_knp
