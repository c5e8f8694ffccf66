"""Generates the committed golden fixtures (small .npz files) from the CPU oracle.

The reference cannot run here (dolfin/petsc4py absent) and holds no golden vectors of its own
(SURVEY.md section 8c), so these fixtures pin the ORACLE (regression pin, MMS-validated) and give the
GPU box reference outputs without needing anything but numpy.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import knpemi_oracle as ko                                    # noqa: E402
from common import synthetic_state, small_3d                   # noqa: E402
from knpemidg.mesh import make_mesh_2D                         # noqa: E402


def one_case(name, mesh_tuple, p=1):
    m, s, f = mesh_tuple
    pb = ko.build_idealized(m, s.array(), f.array(), p=p, membrane_tags=(1,))
    x = synthetic_state(pb)
    A, b_emi, _ = ko.assemble_emi(pb, want_B=False)
    out = dict(degree=np.int64(p), coords=m.coords, cells=m.cells, cell_tags=s.array(), facet_tags=f.array(),
               facet_cells=m.facet_cells, facet_local=m.facet_local,
               x=x, c=pb.c, c_prev=pb.c_prev_n, c_elim=pb.c_elim, phi=pb.phi, phi_M=pb.phi_M,
               I_ch=np.stack([pb.I_ch[i["name"]] for i in pb.ions]),
               kappa=pb.kappa(), emi_Ax=A @ x[0].ravel(), emi_rhs=b_emi,
               knp_Ax=np.stack([ko.assemble_knp(pb, k) @ x[k].ravel() for k in range(pb.N_ions)]),
               knp_rhs=np.stack([ko.knp_rhs(pb, k) for k in range(pb.N_ions)]))
    # one converged splitting step (direct solves) from this state
    import copy
    q = copy.deepcopy(pb)
    E = ko.solve_for_time_step(q, direct=True)
    out.update(step_phi=q.phi - q.phi.mean(), step_c=q.c, step_c_elim=q.c_elim, step_phi_M=q.phi_M,
               step_E=np.stack([E[i["name"]] for i in q.ions]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if k in ("x", "emi_Ax", "step_c")})


if __name__ == "__main__":
    one_case("idealized_2D_r0", make_mesh_2D(0))
    one_case("box_3D_8x4x4", small_3d())
    one_case("box_3D_6x3x3_P2", small_3d((6, 3, 3)), p=2)
    one_case("idealized_2D_r0_P2", make_mesh_2D(0), p=2)
