"""Generates the committed action-potential trajectory fixtures from the CPU oracle ALONE: assembled-CSR forms with
sparse direct solves (oracle/knpemi_oracle.py) + one scipy-LSODA call per membrane facet at the reference's
rtol 1e-8 / atol 0 (oracle/membrane_oracle.py), sequenced as solver.py:1072-1127.  Nothing of the product's
arithmetic is involved (only its mesh generator, whose integer tables are the shared indexing contract).

The GPU tests replay these trajectories two ways (tests/test_gpu_trajectory.py):
  * PDE parity, tight: the stored ODE outputs (phi_M, I_ch_k) are fed to the HIP solver step by step;
  * production run: HIP solver with its own device ODE integrator at the shipped tolerances
    (rtol_emi 1e-5, rtol_knp 1e-7, run_3D.py:172,178) -- errors asserted at EVERY step.

Run:  python tests/golden/make_trajectories.py [name ...]      (minutes: one LU per system per step)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import knpemi_oracle as ko                                    # noqa: E402
import membrane_oracle as mo                                  # noqa: E402
from knpemidg.mesh import make_mesh_3D, make_mesh_2D           # noqa: E402



def _emix_sub():
    from emix_sub import emix_submesh
    return emix_submesh()


STIMULUS = {"stim_amplitude": 10.0}                            # g_syn_bar, run_3D.py:148-153
LOCATOR = lambda x: x[0] < 20.0e-6                             # noqa: E731   run_3D.py:153
N_SAMPLE = 2048


def mean_free(phi, vol):
    return phi - (phi.mean(axis=1) * vol).sum() / vol.sum()


def trajectory(name, mesh_tuple, p, tags_models, n_steps, build=None, stimulus=None, locator=None):
    """tags_models: (membrane tag, True / False = idealized HH with / without stimulus, or a name of membrane_oracle.MODELS)."""
    m, s, f = mesh_tuple
    build = build or ko.build_idealized
    STIMULUS, LOCATOR = stimulus or globals()["STIMULUS"], locator or globals()["LOCATOR"]
    pb = build(m, s.array(), f.array(), p=p, membrane_tags=tuple(t for t, _ in tags_models))
    models = [mo.MembraneOracle(pb, tag, stim is True, pb.C_M, model=None if isinstance(stim, bool) else stim)
              for tag, stim in tags_models]
    E = {ion["name"]: ko.nernst(pb, k) for k, ion in enumerate(pb.ions)}          # solver.py:299-300
    rng = np.random.default_rng(2024)
    sample = np.sort(rng.choice(pb.ndof, size=min(N_SAMPLE, pb.ndof), replace=False))
    nm = len(pb.mem)
    out = dict(degree=np.int64(p), n_steps=np.int64(n_steps), mem=pb.mem, sample=sample,
               ode_phi_M=np.zeros((n_steps, nm)), ode_I_ch=np.zeros((n_steps, len(pb.ions), nm)),
               phi_s=np.zeros((n_steps, len(sample))), c_s=np.zeros((n_steps, pb.N_ions, len(sample))),
               celim_s=np.zeros((n_steps, len(sample))), phi_M=np.zeros((n_steps, nm)),
               E=np.zeros((n_steps, len(pb.ions), nm)), phi_max=np.zeros(n_steps), c_max=np.zeros((n_steps, pb.N_ions)),
               celim_max=np.zeros(n_steps))
    t0 = time.time()
    for k in range(n_steps):
        mo.oracle_membrane_step(pb, E, models, k, pb.dt, STIMULUS, LOCATOR)
        out["ode_phi_M"][k] = pb.phi_M[pb.mem]
        out["ode_I_ch"][k] = np.stack([pb.I_ch[ion["name"]][pb.mem] for ion in pb.ions])
        E = ko.solve_for_time_step(pb, direct=True)
        phi = mean_free(pb.phi, pb.geom.vol)
        out["phi_s"][k] = phi.ravel()[sample]
        out["c_s"][k] = pb.c.reshape(pb.N_ions, -1)[:, sample]
        out["celim_s"][k] = pb.c_elim.ravel()[sample]
        out["phi_M"][k] = pb.phi_M[pb.mem]
        out["E"][k] = np.stack([E[ion["name"]] for ion in pb.ions])
        out["phi_max"][k] = np.abs(phi).max()
        out["c_max"][k] = np.abs(pb.c).reshape(pb.N_ions, -1).max(axis=1)
        out["celim_max"][k] = np.abs(pb.c_elim).max()
        print("%s step %d  phi_M in [%.4f, %.4f]  (%.0f s)" % (name, k, pb.phi_M[pb.mem].min(), pb.phi_M[pb.mem].max(),
                                                               time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


CASES = {
    # BASELINE configs[3] geometry at r=0, P1: 4 axons, two membrane tags, HH with / without stimulus (run_3D.py:196)
    "traj_3D_r0_4axon_P1": lambda: trajectory("traj_3D_r0_4axon_P1", make_mesh_3D(0, n_axons=4), 1, ((1, True), (2, False)), 40),
    # BASELINE configs[2] workload at r=0: the same with Solver(degree_emi=2, degree_knp=2)
    "traj_3D_r0_4axon_P2": lambda: trajectory("traj_3D_r0_4axon_P2", make_mesh_3D(0, n_axons=4), 2, ((1, True), (2, False)), 3),
    # BASELINE configs[0]: 2D neuron r=2, 40 steps
    "traj_2D_r2_P1": lambda: trajectory("traj_2D_r2_P1", make_mesh_2D(2), 1, ((1, True),), 40),
    # BASELINE configs[4] physics on a 17 920-tet piece of its real mesh (tests/emix_sub.py): glial (Kir 4.1) + neuronal (HH, cm / ms / mV)
    # membranes, three subdomains, stimulus g_syn = 5 mS/cm^2 on x < 3e-4 cm (run_EMIx_simulation.py:56-170), 25 steps of 0.1 ms
    "traj_emix_sub_P1": lambda: trajectory("traj_emix_sub_P1", _emix_sub(), 1, ((1, "glial"), (2, "hh_emix")), 25, build=ko.build_emix,
                                           stimulus={"stim_amplitude": 5.0}, locator=lambda x: x[0] < 3.0e-4),
}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(CASES)):
        CASES[name]()
