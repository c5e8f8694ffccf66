"""TEST INFRASTRUCTURE (oracle) -- not part of the product path.

Quadrature rules on reference simplices, returned as barycentric points and
weights that sum to 1 (integral = measure * sum_q w_q f(x_q)).

The reference delegates quadrature to FFC/FIAT (un-vendored third party,
fenics-dolfin 2019.1.x line, unpinned: /root/reference/environment.yml:5).  What
is restated here is the published rule set FIAT's "default" scheme selects
(recalled, third-party): Gauss-Legendre on intervals with (q+2)//2 points; on
triangles the centroid rule (q<=1), Strang-Fix 3/6/6/7/12-point rules for
q = 2..6; collapsed Gauss-Jacobi beyond.  Only NON-polynomial integrands are
sensitive to the rule; on this path those are the membrane terms of L_knp
(estimated degree 5) and ln(c_e/c_i) in the Nernst projection (degree 4),
reference: src/knpemidg/solver.py:603-629, 827-828.  Every other integrand is a
polynomial integrated exactly by any rule of sufficient degree.

PARITY UNPINNED: the reference's tests hold no golden vectors for this path
(SURVEY.md section 8c), so these tables are pinned only by exactness checks.
"""
import numpy as np
from scipy.special import roots_jacobi


def _perm3(a, b):
    # points (a,b,b),(b,a,b),(b,b,a) in barycentric coordinates
    return [[a, b, b], [b, a, b], [b, b, a]]


def _perm6(a, b, c):
    return [[a, b, c], [a, c, b], [b, a, c], [b, c, a], [c, a, b], [c, b, a]]


def _interval(degree):
    n = max(1, (degree + 2) // 2)
    x, w = np.polynomial.legendre.leggauss(n)
    x = 0.5 * (x + 1.0)
    bary = np.stack([1.0 - x, x], axis=1)
    return bary, 0.5 * w


def _triangle(degree):
    if degree <= 1:
        return np.array([[1 / 3, 1 / 3, 1 / 3]]), np.array([1.0])
    if degree == 2:
        return np.array(_perm3(2 / 3, 1 / 6)), np.full(3, 1 / 3)
    if degree == 3:
        pts = _perm6(0.659027622374092, 0.231933368553031, 0.109039009072877)
        return np.array(pts), np.full(6, 1 / 6)
    if degree == 4:
        pts = _perm3(0.816847572980459, 0.091576213509771) + \
            _perm3(0.108103018168070, 0.445948490915965)
        w = [0.109951743655322] * 3 + [0.223381589678011] * 3
        return np.array(pts), np.array(w)
    if degree == 5:
        pts = [[1 / 3, 1 / 3, 1 / 3]] + _perm3(0.797426985353087, 0.101286507323456) + \
            _perm3(0.059715871789770, 0.470142064105115)
        w = [0.225] + [0.125939180544827] * 3 + [0.132394152788506] * 3
        return np.array(pts), np.array(w)
    if degree == 6:
        pts = _perm3(0.873821971016996, 0.063089014491502) + \
            _perm3(0.501426509658179, 0.249286745170910) + \
            _perm6(0.636502499121399, 0.310352451033785, 0.053145049844816)
        w = [0.050844906370207] * 3 + [0.116786275726379] * 3 + [0.082851075618374] * 6
        return np.array(pts), np.array(w)
    return _collapsed(2, degree)


def _collapsed(dim, degree):
    """Stroud conical product of Gauss-Jacobi rules; exact for total degree `degree`."""
    n = degree // 2 + 1
    if dim == 2:
        x0, w0 = roots_jacobi(n, 0, 0)
        x1, w1 = roots_jacobi(n, 1, 0)
        a = 0.5 * (x0 + 1)          # in [0,1]
        b = 0.5 * (x1 + 1)
        A, B = np.meshgrid(a, b, indexing="ij")
        W = np.outer(w0 / 2, w1 / 4)
        # map: x = b? use  lam1 = b, lam2 = a*(1-b), lam0 = 1 - lam1 - lam2
        l1 = B
        l2 = A * (1 - B)
        l0 = 1 - l1 - l2
        bary = np.stack([l0.ravel(), l1.ravel(), l2.ravel()], axis=1)
        w = W.ravel() * 2.0        # reference triangle has area 1/2
        return bary, w
    if dim == 3:
        x0, w0 = roots_jacobi(n, 0, 0)
        x1, w1 = roots_jacobi(n, 1, 0)
        x2, w2 = roots_jacobi(n, 2, 0)
        a = 0.5 * (x0 + 1)
        b = 0.5 * (x1 + 1)
        c = 0.5 * (x2 + 1)
        A, B, C = np.meshgrid(a, b, c, indexing="ij")
        W = w0[:, None, None] / 2 * w1[None, :, None] / 4 * w2[None, None, :] / 8
        l1 = C
        l2 = B * (1 - C)
        l3 = A * (1 - B) * (1 - C)
        l0 = 1 - l1 - l2 - l3
        bary = np.stack([l0.ravel(), l1.ravel(), l2.ravel(), l3.ravel()], axis=1)
        w = W.ravel() * 6.0        # reference tet has volume 1/6
        return bary, w
    raise ValueError(dim)


def simplex_rule(dim, degree):
    """(bary[nq, dim+1], w[nq]) with sum(w) == 1, exact for polynomials of total degree `degree`."""
    degree = int(degree)
    if dim == 0:
        return np.array([[1.0]]), np.array([1.0])
    if dim == 1:
        return _interval(degree)
    if dim == 2:
        return _triangle(degree)
    if dim == 3:
        if degree <= 1:
            return np.array([[0.25] * 4]), np.array([1.0])
        if degree == 2:
            a, b = 0.585410196624969, 0.138196601125011
            pts = [[a, b, b, b], [b, a, b, b], [b, b, a, b], [b, b, b, a]]
            return np.array(pts), np.full(4, 0.25)
        return _collapsed(3, degree)
    raise ValueError(dim)
