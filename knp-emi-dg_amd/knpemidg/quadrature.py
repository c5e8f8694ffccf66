"""Host-side quadrature rules (barycentric points, weights summing to 1).

Used (a) for data terms integrated once on the CPU (manufactured-solution sources and boundary data, SURVEY.md
section 8f-4) and (b) to tabulate the reference basis for the device's DG-p path (knpemidg/dgtab.py ->
knp_set_tabulation).  For (b) the rule matters wherever an integrand is not a polynomial (|.| of the upwind speed, the
rational membrane coefficient, ln(c_e/c_i)): the reference delegates to FIAT's default scheme (un-vendored,
fenics-dolfin 2019.1.x), i.e. Gauss-Legendre with (q+2)//2 points on intervals, the centroid / Strang-Fix 3-, 6-, 6-,
7- and 12-point rules on triangles for q = 1..6, the 4-point rule on tets for q = 2, and the collapsed (Stroud
conical) Gauss-Jacobi product beyond."""
import numpy as np
from scipy.special import roots_jacobi


def _cyc3(a, b):
    return [[a, b, b], [b, a, b], [b, b, a]]


def _all6(a, b, c):
    return [[a, b, c], [a, c, b], [b, a, c], [b, c, a], [c, a, b], [c, b, a]]


_TRI = {
    1: ([[1 / 3, 1 / 3, 1 / 3]], [1.0]),
    2: (_cyc3(2 / 3, 1 / 6), [1 / 3] * 3),
    3: (_all6(0.659027622374092, 0.231933368553031, 0.109039009072877), [1 / 6] * 6),
    4: (_cyc3(0.816847572980459, 0.091576213509771) + _cyc3(0.108103018168070, 0.445948490915965),
        [0.109951743655322] * 3 + [0.223381589678011] * 3),
    5: ([[1 / 3, 1 / 3, 1 / 3]] + _cyc3(0.797426985353087, 0.101286507323456) + _cyc3(0.059715871789770, 0.470142064105115),
        [0.225] + [0.125939180544827] * 3 + [0.132394152788506] * 3),
    6: (_cyc3(0.873821971016996, 0.063089014491502) + _cyc3(0.501426509658179, 0.249286745170910)
        + _all6(0.636502499121399, 0.310352451033785, 0.053145049844816),
        [0.050844906370207] * 3 + [0.116786275726379] * 3 + [0.082851075618374] * 6),
}


def _conical(dim, degree):
    n = degree // 2 + 1
    x0, w0 = roots_jacobi(n, 0, 0)
    a = 0.5 * (x0 + 1)
    x1, w1 = roots_jacobi(n, 1, 0)
    b = 0.5 * (x1 + 1)
    if dim == 2:
        A, B = np.meshgrid(a, b, indexing="ij")
        W = np.outer(w0 / 2, w1 / 4) * 2.0
        l1, l2 = B, A * (1 - B)
        return np.stack([(1 - l1 - l2).ravel(), l1.ravel(), l2.ravel()], axis=1), W.ravel()
    x2, w2 = roots_jacobi(n, 2, 0)
    c = 0.5 * (x2 + 1)
    A, B, C = np.meshgrid(a, b, c, indexing="ij")
    W = (w0[:, None, None] / 2) * (w1[None, :, None] / 4) * (w2[None, None, :] / 8) * 6.0
    l1, l2, l3 = C, B * (1 - C), A * (1 - B) * (1 - C)
    return np.stack([(1 - l1 - l2 - l3).ravel(), l1.ravel(), l2.ravel(), l3.ravel()], axis=1), W.ravel()


def simplex_rule(dim, degree):
    degree = int(degree)
    if dim == 0:
        return np.array([[1.0]]), np.array([1.0])
    if dim == 1:
        x, w = np.polynomial.legendre.leggauss(max(1, (degree + 2) // 2))
        x = 0.5 * (x + 1.0)
        return np.stack([1.0 - x, x], axis=1), 0.5 * w
    if dim == 2:
        if degree <= 6:
            pts, w = _TRI[max(1, degree)]
            return np.array(pts, dtype=np.float64), np.array(w, dtype=np.float64)
        return _conical(2, degree)
    if degree <= 1:
        return np.array([[0.25] * 4]), np.array([1.0])
    if degree == 2:
        a, b = 0.585410196624969, 0.138196601125011
        return np.array([[a, b, b, b], [b, a, b, b], [b, b, a, b], [b, b, b, a]]), np.full(4, 0.25)
    return _conical(3, degree)
