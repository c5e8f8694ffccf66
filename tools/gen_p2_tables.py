"""Generates knp-emi-dg_amd/csrc/p2_tables.hpp: reference-element constants of the matrix-free DG-P2 operator applies
(csrc/apply_p2.hip).  Everything here is geometry independent:

  M3[b][vv']   int phi_b lambda_v lambda_v'   (reference measure 1)  -- exact cell integrals of  coef(P2) * P1 * P1
  MASS[a][b]   int phi_a phi_b
  FMASS[n][n'] int_F psi_n psi_n'             facet P2 mass matrix (membrane coupling of a_emi, solver.py:344)
  facet rules  weights, facet P2 basis psi_n(q) and facet barycentric coordinates lambda_m(q) at the points of the
               rule UFL/FIAT would pick for the dS(0) integrals: degree 3p = 6 for a_emi (all integrands are polynomials of
               degree <= 6, so any exact rule gives the reference's numbers to rounding), degree 3p-1 = 5 for a_knp, whose
               upwind term |D grad(phi).n| is NOT a polynomial -- there the points must be the reference's (FIAT default
               scheme: 7-point Strang-Fix on triangles, 3-point Gauss-Legendre on intervals; knpemidg/quadrature.py)
  FRAME[j]     nibble-packed permutation "facet frame slot -> cell dof" of local facet j:
               slot 0 apex | 1..D facet vertices | then facet edges (m<m') | then apex edges (apex, facet vertex m)

Local dof order of a P2 cell: the D+1 vertices, then edges (a,b), a<b, lexicographic (include/knpemi_hip.h).
Facet vertex m of local facet i is the cell's local vertex m + (m >= i).

Run:  python tools/gen_p2_tables.py   (rewrites the header; committed so that the build needs no Python step)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd"))
from knpemidg.quadrature import simplex_rule          # noqa: E402
from knpemidg.dgtab import tabulate, edges            # noqa: E402


def _num(v, typ, exact):
    if typ != "double":
        return str(int(v))
    if exact:                      # polynomial integrals over the reference simplex are rationals with small denominators
        from fractions import Fraction
        fr = Fraction(float(v)).limit_denominator(1000000)
        assert abs(float(fr) - v) < 1e-13, (v, fr)
        return "0.0" if fr == 0 else "%d.0 / %d.0" % (fr.numerator, fr.denominator)
    return "%.17g" % v


def carr(name, a, typ="double", exact=False):
    a = np.asarray(a)

    def fmt(x):
        if a.ndim == 1:
            return "{" + ", ".join(_num(v, typ, exact) for v in x) + "}"
        return "{" + ",\n        ".join(fmt_sub(r) for r in x) + "}"

    def fmt_sub(r):
        r = np.asarray(r)
        if r.ndim == 1:
            return "{" + ", ".join(_num(v, typ, exact) for v in r) + "}"
        return "{" + ", ".join(fmt_sub(q) for q in r) + "}"
    dims = "".join("[%d]" % n for n in a.shape)
    return "    static constexpr %s %s%s = %s;\n" % (typ, name, dims, fmt(a))


def frame(D, j):
    nv = D + 1
    ed = edges(nv)
    eidx = {e: nv + k for k, e in enumerate(ed)}
    fv = [m + (1 if m >= j else 0) for m in range(D)]
    slots = [j] + fv
    slots += [eidx[(fv[a], fv[b])] for a in range(D) for b in range(a + 1, D)]
    slots += [eidx[tuple(sorted((j, v)))] for v in fv]
    assert sorted(slots) == list(range(nv * (nv + 1) // 2))
    packed = 0
    for s, d in enumerate(slots):
        packed |= d << (4 * s)
    return slots, packed


def tables(D):
    nv = D + 1
    nd = nv * (nv + 1) // 2
    bary, w = simplex_rule(D, 6)
    B, _ = tabulate(2, bary)
    pairs = [(a, b) for a in range(nv) for b in range(a, nv)]
    M3 = np.array([[np.sum(w * B[:, b] * bary[:, v] * bary[:, vp]) for (v, vp) in pairs] for b in range(nd)])
    MASS = np.einsum("q,qa,qb->ab", w, B, B)
    mu, wf = simplex_rule(D - 1, 6)
    Bf, _ = tabulate(2, mu)
    FMASS = np.einsum("q,qa,qb->ab", wf, Bf, Bf)
    out = {"M3": M3, "MASS": MASS, "FMASS": FMASS}
    for tag, deg in (("E", 6), ("K", 5)):
        mu, wq = simplex_rule(D - 1, deg)
        psi, _ = tabulate(2, mu)
        out["W" + tag] = wq
        out["PSI" + tag] = psi
        out["LAM" + tag] = mu
    out["FRAME_SLOTS"] = np.array([frame(D, j)[0] for j in range(nv)])
    out["FRAME_PACKED"] = [frame(D, j)[1] for j in range(nv)]
    return out


def mono_tables():
    """Tables of the MONOMIAL-PRODUCT form of the facet integrals of a_emi on triangles (csrc/apply_p2.hip, round 4): every integrand
    is a polynomial, so instead of sampling at the 12 points of the degree-6 rule the kernel multiplies the P2 traces as polynomials in
    the facet's barycentric coordinates and integrates the product exactly.
      degree-2 monomials (order of a P2 trace after conversion):  l0^2 l1^2 l2^2 l0l1 l0l2 l1l2
      IDX22[i][j]  degree-4 monomial of (degree-2 monomial i) * (degree-2 monomial j)
      IDX21[i][m]  degree-3 monomial of (degree-2 monomial i) * l_m
      IDX31[k][m]  degree-4 monomial of (degree-3 monomial k) * l_m          (multiplication by 1 = l0 + l1 + l2)
      I4[a][n]     int l^a psi_n   (psi_n: the facet P2 Lagrange basis, frame order v0 v1 v2 e01 e02 e12; reference measure 1)
      I4L[a][m]    int l^a l_m"""
    import itertools
    from math import factorial
    M1 = [(1, 0, 0), (0, 1, 0), (0, 0, 1)]
    M2 = [(2, 0, 0), (0, 2, 0), (0, 0, 2), (1, 1, 0), (1, 0, 1), (0, 1, 1)]

    def mons(d):
        return [a for a in itertools.product(range(d + 1), repeat=3) if sum(a) == d]
    M3, M4 = mons(3), mons(4)
    i3 = {a: k for k, a in enumerate(M3)}
    i4 = {a: k for k, a in enumerate(M4)}
    add = lambda a, b: tuple(x + y for x, y in zip(a, b))
    integ = lambda a: 2.0 * factorial(a[0]) * factorial(a[1]) * factorial(a[2]) / factorial(sum(a) + 2)
    psi = np.zeros((6, 6))                       # Lagrange P2 in degree-2 monomials (1 = l0 + l1 + l2 folded in)
    for v in range(3):
        psi[v, v] = 1.0
        for k, (a, b) in enumerate([(0, 1), (0, 2), (1, 2)]):
            if v in (a, b):
                psi[v, 3 + k] = -1.0
    for k in range(3):
        psi[3 + k, 3 + k] = 4.0
    out = {"IDX22": np.array([[i4[add(a, b)] for b in M2] for a in M2]),
           "IDX21": np.array([[i3[add(a, m)] for m in M1] for a in M2]),
           "IDX31": np.array([[i4[add(a, m)] for m in M1] for a in M3]),
           "I4": np.array([[sum(psi[n, b] * integ(add(al, M2[b])) for b in range(6)) for n in range(6)] for al in M4]),
           "I4L": np.array([[integ(add(al, m)) for m in M1] for al in M4])}
    return out


def write_mono():
    t = mono_tables()
    s = ("// GENERATED by tools/gen_p2_tables.py -- do not edit.  Monomial-product form of the triangle-facet integrals of a_emi (DG-P2, 3D):\n"
         "// index maps of the polynomial products and the exact integrals of degree-4 monomials against the facet bases; see the generator.\n"
         "#pragma once\n\nstruct P2Mono {\n    static constexpr int N2 = 6, N3 = 10, N4 = 15;\n")
    for k in ("IDX22", "IDX21", "IDX31"):
        s += carr(k, t[k], "int")
    for k in ("I4", "I4L"):
        s += carr(k, t[k], exact=True)
    s += "};\n"
    path = os.path.join(ROOT, "knp-emi-dg_amd", "csrc", "p2_mono_tables.hpp")
    with open(path, "w") as f:
        f.write(s)
    print("wrote", path)


def main():
    write_mono()
    s = ("// GENERATED by tools/gen_p2_tables.py -- do not edit.  Reference-element constants of the matrix-free DG-P2 applies\n"
         "// (apply_p2.hip); see the generator for definitions.\n#pragma once\n#include <cstdint>\n\n"
         "template <int D> struct P2Tab;\n\n")
    for D in (2, 3):
        t = tables(D)
        nv = D + 1
        s += "template <> struct P2Tab<%d> {\n" % D
        s += "    static constexpr int NV = %d, ND = %d, NFV = %d, NF = %d, NQE = %d, NQK = %d, NPAIR = %d;\n" % (
            nv, nv * (nv + 1) // 2, D, D * (D + 1) // 2, len(t["WE"]), len(t["WK"]), nv * (nv + 1) // 2)
        for k in ("M3", "MASS", "FMASS", "WE", "PSIE", "LAME", "WK", "PSIK", "LAMK"):
            s += carr(k, t[k], exact=k in ("M3", "MASS", "FMASS"))
        s += carr("FRAME_SLOTS", t["FRAME_SLOTS"], "int")
        s += "    static constexpr uint64_t FRAME_PACKED[%d] = {%s};\n" % (nv, ", ".join("0x%xull" % v for v in t["FRAME_PACKED"]))
        s += "};\n\n"
    path = os.path.join(ROOT, "knp-emi-dg_amd", "csrc", "p2_tables.hpp")
    with open(path, "w") as f:
        f.write(s)
    print("wrote", path)


if __name__ == "__main__":
    main()
