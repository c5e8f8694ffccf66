"""Reference-basis tabulations for the device's DG-p path (csrc/tab_dg.hip, knp_set_tabulation).

Lagrange P2 on simplices in barycentric coordinates; local dof order: the dim+1 vertices, then the edge midpoints
(a, b), a < b, in lexicographic order.  The quadrature degree of every integral class is the one UFL estimates for the
corresponding reference form (forms: src/knpemidg/solver.py:270-403, 534-663, 808-845)."""
import numpy as np

from knpemidg.quadrature import simplex_rule

# slot ids == enum knp_tab_slot (include/knpemi_hip.h)
CELL_STIFF, CELL_RHS_EMI, CELL_MASS, FACET_EMI, FACET_MEM, FACET_KNP, FACET_RHS_EMI, FACET_MEM_LIN, FACET_MEM_KNP, \
    FACET_AVG, FACET_NERNST = range(11)


def slot_degrees(p):
    return {
        CELL_STIFF: max(2, 3 * p - 2, 2 * p),
        CELL_RHS_EMI: max(1, 2 * p - 2),
        CELL_MASS: 2 * p,
        FACET_EMI: 3 * p,
        FACET_MEM: 2 * p,
        FACET_KNP: max(2, 3 * p - 1),
        FACET_RHS_EMI: max(1, 2 * p - 1),
        FACET_MEM_LIN: max(1, 2 * p - 1),
        FACET_MEM_KNP: 5 * p,
        FACET_AVG: max(1, p),
        FACET_NERNST: 2 * p + 2,
    }


def edges(nv):
    return [(a, b) for a in range(nv) for b in range(a + 1, nv)]


def tabulate(p, bary):
    """B[q, j] and dB[q, j, l] = d phi_j / d lambda_l at barycentric points bary[q, :]."""
    bary = np.asarray(bary, dtype=np.float64)
    nq, nv = bary.shape
    if p == 1:
        return bary.copy(), np.broadcast_to(np.eye(nv), (nq, nv, nv)).copy()
    if p != 2:
        raise ValueError("degree 1 or 2")
    ed = edges(nv)
    B = np.zeros((nq, nv + len(ed)))
    dB = np.zeros((nq, nv + len(ed), nv))
    for a in range(nv):
        B[:, a] = bary[:, a] * (2.0 * bary[:, a] - 1.0)
        dB[:, a, a] = 4.0 * bary[:, a] - 1.0
    for e, (a, b) in enumerate(ed):
        B[:, nv + e] = 4.0 * bary[:, a] * bary[:, b]
        dB[:, nv + e, a] = 4.0 * bary[:, b]
        dB[:, nv + e, b] = 4.0 * bary[:, a]
    return B, dB


def tables(dim, p):
    """{slot: (nloc, nq, w, B[nloc, nq, nd], dB[nloc, nq, nd, dim+1])}."""
    out = {}
    for slot, deg in slot_degrees(p).items():
        if slot < FACET_EMI:
            bary, w = simplex_rule(dim, deg)
            B, dB = tabulate(p, bary)
            out[slot] = (1, len(w), np.ascontiguousarray(w), np.ascontiguousarray(B[None]), np.ascontiguousarray(dB[None]))
        else:
            mu, w = simplex_rule(dim - 1, deg)
            Bs, dBs = [], []
            for i in range(dim + 1):
                # facet vertex m is the cell's local vertex m + (m >= i): insert a zero at position i
                B, dB = tabulate(p, np.insert(mu, i, 0.0, axis=1))
                Bs.append(B)
                dBs.append(dB)
            out[slot] = (dim + 1, len(w), np.ascontiguousarray(w), np.ascontiguousarray(np.array(Bs)),
                         np.ascontiguousarray(np.array(dBs)))
    return out
