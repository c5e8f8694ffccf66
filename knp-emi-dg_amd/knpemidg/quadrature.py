"""Host-side quadrature for data terms that are integrated once on the CPU (manufactured-solution sources and
boundary data): Gauss-Legendre on intervals, collapsed Gauss-Jacobi on triangles / tets.
Barycentric points, weights summing to 1."""
import numpy as np
from scipy.special import roots_jacobi


def simplex_rule(dim, degree):
    n = max(1, degree // 2 + 1)
    if dim == 0:
        return np.array([[1.0]]), np.array([1.0])
    x0, w0 = roots_jacobi(n, 0, 0)
    a = 0.5 * (x0 + 1)
    if dim == 1:
        return np.stack([1 - a, a], axis=1), 0.5 * w0
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
