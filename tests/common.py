"""Shared helpers for parity tests: seeded synthetic inputs of BASELINE config 2."""
import numpy as np

import knpemi_oracle as ko


def synthetic_state(pb, seed=0, volt=1.0):
    """x ~ U(-1,1) seed 0; c_k = tag-wise ICs * (1 + 0.01 U(-1,1)) seed 1; phi ~ 0.07 U(-1,1) seed 2
    (BASELINE.md section 4, config 2).  Also perturbs phi_M and channel currents.
    volt = 1e3 for configurations in mV (EMIx)."""
    nc, nd = pb.mesh.num_cells(), pb.nd
    x = np.random.default_rng(seed).uniform(-1, 1, size=(pb.N_ions, nc, nd))
    r1 = np.random.default_rng(seed + 1)
    pb.c = pb.c * (1 + 0.01 * r1.uniform(-1, 1, size=pb.c.shape))
    pb.c_elim = pb.c_elim * (1 + 0.01 * r1.uniform(-1, 1, size=pb.c_elim.shape))
    pb.c_prev_n = pb.c * (1 + 0.001 * r1.uniform(-1, 1, size=pb.c.shape))
    pb.phi = 0.07 * volt * np.random.default_rng(seed + 2).uniform(-1, 1, size=(nc, nd))
    r3 = np.random.default_rng(seed + 3)
    nf = pb.mesh.num_facets()
    pb.phi_M = np.zeros(nf)
    pb.phi_M[pb.mem] = volt * (-0.07 + 0.01 * r3.uniform(-1, 1, size=len(pb.mem)))
    for name in pb.I_ch:
        pb.I_ch[name] = np.zeros(nf)
        pb.I_ch[name][pb.mem] = 1e-3 * volt * r3.uniform(-1, 1, size=len(pb.mem))
    return x


def device_for(pb, **kw):
    from knpemidg import _abi
    dev = _abi.Device(pb.mesh, pb.cell_tags, pb.facet_tags, pb.membrane_tags, len(pb.ions), degree=pb.p, **kw)
    z = [ion["z"] for ion in pb.ions]
    D = np.stack([ion["D"] for ion in pb.ions])
    dev.set_params(pb.C_M, pb.dt, pb.F, pb.R, pb.T, pb.C_phi, pb.tau, pb.tau, z, D, rho=pb.rho,
                   splitting=pb.splitting)
    return dev


def push_state(dev, pb):
    from knpemidg import _abi as A
    dev.upload(A.F_C, pb.c)
    dev.upload(A.F_C_PREV, pb.c_prev_n)
    dev.upload(A.F_C_ELIM, pb.c_elim)
    dev.upload(A.F_PHI, pb.phi)
    dev.upload(A.F_PHI_M, pb.phi_M)
    dev.upload(A.F_I_CH, np.stack([pb.I_ch[ion["name"]] for ion in pb.ions]))


def relerr(a, b):
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def small_3d(n=(8, 4, 4)):
    """Tiny 3D box with one ICS block: fast enough for the oracle's direct solves."""
    from knpemidg.mesh import BoxMesh, MeshFunction, _tag_box
    nx, ny, nz = n
    mesh = BoxMesh((0, 0, 0), (nx * 1.0, ny * 0.1, nz * 0.1), nx, ny, nz)
    sub = MeshFunction(mesh, 3, 0)
    surf = MeshFunction(mesh, 2, 0)
    _tag_box(mesh, sub, surf, (2, 0.1, 0.1), (nx - 2, (ny - 1) * 0.1, (nz - 1) * 0.1), 1)
    surf.array()[mesh.exterior_facets()] = 5
    mesh.coords *= 1e-6
    return mesh, sub, surf


def mean_free(phi, vol):
    """Subtract the volume-weighted mean (phi is defined modulo a constant, solver.py:464-490)."""
    phi = np.asarray(phi).reshape(len(vol), -1)
    mean = (phi.mean(axis=1) * vol).sum() / vol.sum()
    return phi - mean


def tortuosity_3d(resolution=0, p=1):
    """The 4-axon idealized mesh with the coefficients of run_tortuosity.py (oracle.build_tortuosity): coordinates in cm, the cells
    of the three tag-2 axons relabelled as subdomain 2 (glial), so that the structured kernels see three materials, a non-zero
    background charge rho_sub and an eliminated ion with z = -1."""
    from knpemidg.mesh import make_mesh_3D, Mesh
    m, s, f = make_mesh_3D(resolution)
    sub = s.array().astype(np.int64).copy()
    mid = m.coords[m.cells].mean(axis=1)
    sub[(sub == 1) & ((mid[:, 1] > 0.45e-6) | (mid[:, 2] > 0.45e-6))] = 2
    mesh = Mesh(m.coords * 100.0, m.cells)
    return ko.build_tortuosity(mesh, sub, f.array(), p=p)
