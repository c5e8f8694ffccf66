"""Reference (numpy) V-cycle over a knpemidg.amg hierarchy -- test infrastructure mirroring csrc/amg.hip."""
import numpy as np


def cheb_smooth(lv, x, b, zero_guess):
    lmax, lmin = lv.rho, lv.cheb_lower * lv.rho
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    rho = 1.0 / sigma
    r = b.copy() if zero_guess else b - lv.A @ x
    d = lv.dinv * r / theta
    for k in range(lv.cheb_degree):
        x = x + d
        if k == lv.cheb_degree - 1:
            break
        r = r - lv.A @ d
        rho_new = 1.0 / (2.0 * sigma - rho)
        d = rho_new * rho * d + (2.0 * rho_new / delta) * (lv.dinv * r)
        rho = rho_new
    return x


def vcycle(levels, b, l=0):
    lv = levels[l]
    if l == len(levels) - 1:
        return lv.pinv @ b
    x = cheb_smooth(lv, np.zeros_like(b), b, True)
    r = b - lv.A @ x
    x = x + lv.P @ vcycle(levels, lv.R @ r, l + 1)
    return cheb_smooth(lv, x, b, False)
