"""Manufactured solution in time (reference: tests/mms_time.py:6-164): fields linear in space (so P1 is exact in
space) with trigonometric / quadratic time dependence.  Every callable f(X) evaluates at the CURRENT value of the
shared time Constant `t` -- the same object the driver hands to solve_system_passive, exactly as the reference's UFL
expressions hold `t` (the data of step k is therefore taken at t_k, solver.py:845 advances t at the end of the step)."""
from collections import namedtuple

import numpy as np
import sympy as sy

MMSData = namedtuple('MMSData', ('solution', 'rhs', 'normals', 'time_dependent'))


def setup_mms(params, t, mesh=None):
    x, y, ts = sy.symbols("x y t")
    pi = sy.pi
    P = params
    D = {"a1": P.D_a1, "a2": P.D_a2, "b1": P.D_b1, "b2": P.D_b2, "c1": P.D_c1, "c2": P.D_c2}
    Cc = {"a1": P.C_a1, "a2": P.C_a2, "b1": P.C_b1, "b2": P.C_b2, "c1": P.C_c1, "c2": P.C_c2}
    z = {"a": P.z_a, "b": P.z_b, "c": P.z_c}
    F, psi, C_phi = P.F, P.F / (P.R * P.temperature), P.C_phi
    k = {}
    k["a1"] = 1 + (x + y) + 0.2 * sy.cos(2 * pi * ts)
    k["b1"] = 1 + (x + y) + 0.3 * sy.cos(2 * pi * ts)
    k["c1"] = -1 / z["c"] * (z["a"] * k["a1"] + z["b"] * k["b1"])
    k["a2"] = 1 + (x + y) + 0.5 * sy.sin(2 * pi * ts)
    k["b2"] = 1 + (x + y) + 0.6 * sy.sin(2 * pi * ts)
    k["c2"] = -1 / z["c"] * (z["a"] * k["a2"] + z["b"] * k["b2"])
    phi = {"1": (1 + x + y) * (1 + ts ** 2), "2": (1 + x - y) * (1 + ts ** 2)}

    grad = lambda f: sy.Matrix([sy.diff(f, x), sy.diff(f, y)])
    div = lambda v: sy.diff(v[0], x) + sy.diff(v[1], y)
    J = {key: -D[key] * grad(k[key]) - z[key[0]] * D[key] * psi * k[key] * grad(phi[key[1]]) for key in k}
    f_k = {key: sy.diff(k[key], ts) + div(J[key]) for key in k}
    f_phi = {dom: F * sum(z[s] * div(J[s + dom]) for s in "abc") for dom in "12"}

    def lam(e, at=None):
        f = sy.lambdify((x, y, ts), e, "numpy")
        return lambda X: np.broadcast_to(np.asarray(f(X[..., 0], X[..., 1], float(t) if at is None else at), dtype=np.float64),
                                         X.shape[:-1])

    def lamv(v):
        f0, f1 = lam(v[0]), lam(v[1])
        return lambda X: np.stack([f0(X), f1(X)], axis=-1)

    normals = {1: (-1, 0), 2: (0, -1), 3: (1, 0), 4: (0, 1)}
    g_phi, g_stress, g_rob = {}, {}, {}
    for tag, n1 in normals.items():
        nv = sy.Matrix(n1)
        dn = lambda v: (v.T * nv)[0]
        g_phi[tag] = lam(phi["1"] - phi["2"] - (1 / C_phi) * F * sum(z[s] * dn(J[s + "1"]) for s in "abc"))
        g_stress[tag] = lam(-F * sum(z[s] * (dn(J[s + "1"]) - dn(J[s + "2"])) for s in "abc"))
        for key in k:
            g_rob.setdefault(key, {})[tag] = lam(phi["1"] - phi["2"] - (1 / Cc[key]) * dn(J[key]))
    solution = {"c_" + key: lam(v) for key, v in k.items()}
    solution.update({"phi_1": lam(phi["1"]), "phi_2": lam(phi["2"])})
    solution.update({"c_%s_init" % key: lam(v, at=0.0) for key, v in k.items()})
    rhs = {"volume_phi_1": lam(f_phi["1"]), "volume_phi_2": lam(f_phi["2"])}
    rhs.update({"volume_c_" + key: lam(v) for key, v in f_k.items()})
    rhs["bdry"] = {"neumann_a": lamv(J["a2"]), "neumann_b": lamv(J["b2"]), "neumann_c": lamv(J["c2"]),
                   "stress": g_stress, "u_phi": g_phi}
    rhs["bdry"].update({"u_" + key: g_rob[key] for key in k})
    return MMSData(solution=solution, rhs=rhs, normals=normals, time_dependent=True)
