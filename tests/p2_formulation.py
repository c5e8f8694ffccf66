"""numpy restatement of the FORMULATION the matrix-free DG-P2 kernels use (csrc/apply_p2.hip): exact cell integrals through the
reference tensor M3, facet integrals in the "facet frame" with the generated tables of csrc/p2_tables.hpp, geometry in Gram
form.  Test infrastructure: lets the formulation (and the generated tables) be checked against the oracle's assembled
matrices on the CPU, before and independently of the HIP code.  Vectorised over cells; not fast, not shipped."""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HDR = os.path.join(os.path.dirname(HERE), "knp-emi-dg_amd", "csrc", "p2_tables.hpp")


def load_tables(D):
    """Parse the generated C++ header (so that the test checks the numbers the kernels compile in)."""
    txt = open(HDR).read()
    body = txt[txt.index("struct P2Tab<%d>" % D):]
    body = body[:body.index("};\n\n") + 2]
    out = {}
    for m in re.finditer(r"static constexpr (double|int) (\w+)((?:\[\d+\])+) = (\{.*?\});", body, re.S):
        typ, name, dims, val = m.groups()
        shape = [int(v) for v in re.findall(r"\[(\d+)\]", dims)]
        val = val.replace("{", "[").replace("}", "]")
        out[name] = np.array(eval(val), dtype=np.float64 if typ == "double" else np.int64).reshape(shape)
    m = re.search(r"FRAME_PACKED\[\d+\] = \{(.*?)\}", body)
    out["FRAME_PACKED"] = [int(v.strip().rstrip("ull"), 16) for v in m.group(1).split(",")]
    return out


class Geo:
    """Per-cell Gram-form geometry + per-(cell, local facet) neighbour data, from the mesh tables."""

    def __init__(self, mesh, cell_tags, facet_tags, membrane_tags):
        D = mesh.gdim
        nv = D + 1
        nc = mesh.num_cells()
        X = mesh.coords[mesh.cells]
        J = (X[:, 1:, :] - X[:, :1, :]).transpose(0, 2, 1)
        Jinv = np.linalg.inv(J)
        g = np.empty((nc, nv, D))
        g[:, 1:, :] = Jinv
        g[:, 0, :] = -Jinv.sum(axis=1)
        self.G = np.einsum("cad,cbd->cab", g, g)
        self.vol = np.abs(np.linalg.det(J)) / (2.0 if D == 2 else 6.0)
        e = X[:, :, None, :] - X[:, None, :, :]
        h = np.sqrt((e ** 2).sum(axis=3).max(axis=(1, 2)))
        cf = mesh.cell_facets
        fc, fl = mesh.facet_cells, mesh.facet_local.astype(np.int64)
        side = (fc[cf, 0] != np.arange(nc)[:, None]).astype(np.int64)
        self.nb = np.take_along_axis(fc[cf], (1 - side)[:, :, None], axis=2)[:, :, 0]
        self.nj = np.take_along_axis(fl[cf], (1 - side)[:, :, None], axis=2)[:, :, 0]
        has = self.nb >= 0
        ft = np.asarray(facet_tags)[cf]
        self.kind = np.where(~has, 2, np.where(ft == 0, 0, np.where(np.isin(ft, list(membrane_tags)), 1, 3)))
        apex = mesh.coords[mesh.cells[np.maximum(self.nb, 0), np.maximum(self.nj, 0)]] - X[:, :1, :]
        self.L = np.einsum("cad,cid->cia", g, apex)
        self.L[:, :, 0] += 1.0
        self.sqG = np.sqrt(np.einsum("cii->ci", self.G))
        self.hinv = np.where(has, 2.0 / (h[:, None] + h[np.maximum(self.nb, 0)]), 0.0)
        self.D, self.nv, self.nc = D, nv, nc


def _edge_index(nv):
    idx = {}
    k = nv
    for a in range(nv):
        for b in range(a + 1, nv):
            idx[(a, b)] = idx[(b, a)] = k
            k += 1
    return idx


def _nodal_gradients(x, nv, eidx):
    """U[c, l, v] = d u / d lambda_l at vertex v (P2 nodal values x[c, nd])."""
    nc = x.shape[0]
    U = np.zeros((nc, nv, nv))
    for v in range(nv):
        for l in range(nv):
            U[:, l, v] = 3.0 * x[:, v] if l == v else 4.0 * x[:, eidx[(v, l)]] - x[:, l]
    return U


def _project(H, nv, eidx, nd):
    """y[c, a] = sum_{l, v} H[c, l, v] * d phi_a / d lambda_l (v)."""
    y = np.zeros((H.shape[0], nd))
    for a in range(nv):
        y[:, a] = 3.0 * H[:, a, a] - sum(H[:, a, v] for v in range(nv) if v != a)
    for (a, b), k in eidx.items():
        if a < b:
            y[:, k] = 4.0 * (H[:, a, b] + H[:, b, a])
    return y


def _pair_matrix(T, coef, nv):
    """W[c, v, v'] = sum_b coef[c, b] M3[b, pair(v, v')]."""
    Wp = coef @ T["M3"]
    W = np.zeros((coef.shape[0], nv, nv))
    k = 0
    for a in range(nv):
        for b in range(a, nv):
            W[:, a, b] = W[:, b, a] = Wp[:, k]
            k += 1
    return W


def _frames(T, x, j):
    """x in the facet frame of (runtime) local facet j[c]:  x[c, FRAME_SLOTS[j[c], s]]."""
    return np.take_along_axis(x, T["FRAME_SLOTS"][j], axis=1)


def _dn_vertices(F, gnA, gnV, D, nfe):
    """normal derivative at the D facet vertices from a frame vector F [apex | fv | fe | ae]."""
    fe = {}
    k = 1 + D
    for a in range(D):
        for b in range(a + 1, D):
            fe[(a, b)] = fe[(b, a)] = k
            k += 1
    out = np.zeros((F.shape[0], D))
    for m in range(D):
        s = 3.0 * gnV[:, m] * F[:, 1 + m] + gnA * (4.0 * F[:, 1 + D + nfe + m] - F[:, 0])
        for mp in range(D):
            if mp != m:
                s = s + gnV[:, mp] * (4.0 * F[:, fe[(m, mp)]] - F[:, 1 + mp])
        out[:, m] = s
    return out, fe


def _back_project(Y, Tm, gnA, gnV, D, nfe, fe):
    """Y (frame) += sum_m T_m * (normal derivative of the frame basis functions at facet vertex m)."""
    for m in range(D):
        Y[:, 1 + m] += 3.0 * gnV[:, m] * Tm[:, m]
        Y[:, 0] += -gnA * Tm[:, m]
        Y[:, 1 + D + nfe + m] += 4.0 * gnA * Tm[:, m]
        for mp in range(D):
            if mp != m:
                Y[:, 1 + mp] += -gnV[:, mp] * Tm[:, m]
                Y[:, fe[(m, mp)]] += 4.0 * gnV[:, mp] * Tm[:, m]


def emi_apply(geo, T, x, kappa, tau, C_phi):
    D, nv, nc = geo.D, geo.nv, geo.nc
    nd = nv * (nv + 1) // 2
    nfe = D * (D - 1) // 2
    nf = D + nfe
    eidx = _edge_index(nv)
    x = x.reshape(nc, nd)
    kappa = kappa.reshape(nc, nd)
    # cells: vol * sum G_ll' int kappa d_l u d_l' v
    U = _nodal_gradients(x, nv, eidx)
    W = _pair_matrix(T, kappa, nv)
    Z = np.einsum("cvw,clv->clw", W, U)
    H = np.einsum("ckl,clw->ckw", geo.G, Z) * geo.vol[:, None, None]
    y = _project(H, nv, eidx, nd)
    for i in range(nv):
        kind = geo.kind[:, i]
        act = np.nonzero(kind <= 1)[0]
        if not len(act):
            continue
        nb, j = geo.nb[act, i], geo.nj[act, i]
        Fo = x[act][:, T["FRAME_SLOTS"][i]]
        Ko = kappa[act][:, T["FRAME_SLOTS"][i]]
        Fn = _frames(T, x[nb], j)
        Kn = _frames(T, kappa[nb], j)
        ju = Fo[:, 1:1 + nf] - Fn[:, 1:1 + nf]
        sqG = geo.sqG[act, i]
        area = sqG * D * geo.vol[act]
        Y = np.zeros((len(act), nd))
        mem = kind[act] == 1
        # membrane: C_phi int jump(u) v
        Y[mem, 1:1 + nf] = (C_phi * area[mem])[:, None] * (ju[mem] @ T["FMASS"].T)
        sip = ~mem
        fv = [m + (1 if m >= i else 0) for m in range(D)]
        gnA = -sqG
        gnV = -geo.G[act][:, fv, i] / sqG[:, None]
        Li = geo.L[act, i]
        gr = gnA / Li[:, i]
        gnVn = gnV - Li[:, fv] * gr[:, None]
        dno, fe = _dn_vertices(Fo, gnA, gnV, D, nfe)
        dnn, _ = _dn_vertices(Fn, gr, gnVn, D, nfe)
        pen = tau * geo.hinv[act, i]
        r = np.zeros((len(act), nf))
        Tm = np.zeros((len(act), D))
        for q in range(len(T["WE"])):
            psi, lam, w = T["PSIE"][q], T["LAME"][q], T["WE"][q]
            ko, kn, jq = Ko[:, 1:1 + nf] @ psi, Kn[:, 1:1 + nf] @ psi, ju @ psi
            flux = -0.5 * (ko * (dno @ lam) + kn * (dnn @ lam)) + pen * 0.5 * (ko + kn) * jq
            t = -0.5 * ko * jq
            r += w * flux[:, None] * psi[None, :]
            Tm += w * t[:, None] * lam[None, :]
        Ys = np.zeros((len(act), nd))
        Ys[:, 1:1 + nf] = area[:, None] * r
        _back_project(Ys, area[:, None] * Tm, gnA, gnV, D, nfe, fe)
        Y[sip] = Ys[sip]
        np.add.at(y, (act[:, None], T["FRAME_SLOTS"][i][None, :]), Y)
    return y.ravel()


def knp_apply(geo, T, x, phi, Dk, z, psi_c, tau, dt):
    """One species: x [nc*nd], Dk [nc] (cell-wise diffusion coefficient), z valence."""
    D, nv, nc = geo.D, geo.nv, geo.nc
    nd = nv * (nv + 1) // 2
    nfe = D * (D - 1) // 2
    nf = D + nfe
    eidx = _edge_index(nv)
    x = x.reshape(nc, nd)
    phi = phi.reshape(nc, nd)
    zp = z * psi_c
    # cells: 1/dt M + D K + z psi D int u grad(phi).grad(v)
    y = (geo.vol / dt)[:, None] * (x @ T["MASS"].T)
    U = _nodal_gradients(x, nv, eidx)
    m1 = 1.0 / ((D + 1) * (D + 2))
    Z = m1 * (U + U.sum(axis=2, keepdims=True))
    P = _nodal_gradients(phi, nv, eidx)
    Wg = np.einsum("ckl,clv->ckv", geo.G, P)                    # grad(phi).grad(lambda_k) at vertex v
    Q = _pair_matrix(T, x, nv)
    R = np.einsum("ckv,cvw->ckw", Wg, Q)
    H = (np.einsum("ckl,clw->ckw", geo.G, Z) + zp * R) * (Dk * geo.vol)[:, None, None]
    y += _project(H, nv, eidx, nd)
    for i in range(nv):
        act = np.nonzero(geo.kind[:, i] == 0)[0]
        if not len(act):
            continue
        nb, j = geo.nb[act, i], geo.nj[act, i]
        Fo, Po = x[act][:, T["FRAME_SLOTS"][i]], phi[act][:, T["FRAME_SLOTS"][i]]
        Fn, Pn = _frames(T, x[nb], j), _frames(T, phi[nb], j)
        Dc, D2 = Dk[act], Dk[nb]
        sqG = geo.sqG[act, i]
        area = sqG * D * geo.vol[act]
        fv = [m + (1 if m >= i else 0) for m in range(D)]
        gnA = -sqG
        gnV = -geo.G[act][:, fv, i] / sqG[:, None]
        Li = geo.L[act, i]
        gr = gnA / Li[:, i]
        gnVn = gnV - Li[:, fv] * gr[:, None]
        dno, fe = _dn_vertices(Fo, gnA, gnV, D, nfe)
        dnn, _ = _dn_vertices(Fn, gr, gnVn, D, nfe)
        dpo, _ = _dn_vertices(Po, gnA, gnV, D, nfe)
        dpn, _ = _dn_vertices(Pn, gr, gnVn, D, nfe)
        pen = tau * geo.hinv[act, i]
        r = np.zeros((len(act), nf))
        Tm = np.zeros((len(act), D))
        for q in range(len(T["WK"])):
            psi, lam, w = T["PSIK"][q], T["LAMK"][q], T["WK"][q]
            uo, un = Fo[:, 1:1 + nf] @ psi, Fn[:, 1:1 + nf] @ psi
            sp = Dc * (dpo @ lam)
            sm = -D2 * (dpn @ lam)
            upo, upn = 0.5 * (sp + np.abs(sp)), 0.5 * (sm + np.abs(sm))
            flux = -0.5 * (Dc * (dno @ lam) + D2 * (dnn @ lam)) + pen * (Dc * uo - D2 * un) - zp * (upo * uo - upn * un)
            t = -0.5 * Dc * (uo - un)
            r += w * flux[:, None] * psi[None, :]
            Tm += w * t[:, None] * lam[None, :]
        Y = np.zeros((len(act), nd))
        Y[:, 1:1 + nf] = area[:, None] * r
        _back_project(Y, area[:, None] * Tm, gnA, gnV, D, nfe, fe)
        np.add.at(y, (act[:, None], T["FRAME_SLOTS"][i][None, :]), Y)
    return y.ravel()
