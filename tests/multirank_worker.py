"""One rank of tests/test_gpu_multirank.py: the partitioned idealized 3D solver on the shared-memory communicator (several ranks on
one GPU), a few stimulated steps at tight tolerances; writes its owned cells' results for the parent to compare with the single-rank run.
usage: multirank_worker.py rank world shm_name outdir method n_axons steps"""
import os
import sys

import numpy as np

rank, world, name, outdir, method, n_axons, steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "examples", "idealized_geometries")]
os.environ["WORLD_SIZE"] = str(world)
os.environ["KNP_COMM_SHM"] = name
from idealized_common import SolverIdealized, physical_setup, solver_parameters, Constant          # noqa: E402
from knpemidg.mesh import make_mesh_3D                                                              # noqa: E402
from knpemidg.models import mm_hh, mm_hh_no_stim                                                    # noqa: E402
from knpemidg.partition import distribute_solver                                                    # noqa: E402

if method == "emix":
    # BASELINE configs[4]: the EMIx tissue reconstruction (unstructured, glial + neuronal membranes), RCB partition
    sys.path.insert(0, os.path.join(ROOT, "examples", "emix_simulations"))
    import emix_common
    S = emix_common.make_distributed_solver(rank, world, 0, None)
    S._unpack_solver_params(emix_common.solver_parameters()._replace(rtol_emi=1e-10, rtol_knp=1e-12))
else:
    mesh_tuple = make_mesh_3D(0, n_axons=n_axons)
    ode_models = {1: mm_hh, 2: mm_hh_no_stim} if n_axons > 1 else {1: mm_hh}
    params, ion_list, stim_params = physical_setup(1.0e-4)
    # "thin": three slabs, the middle one 2 % of the cells -- every cell of that rank touches a cut (n_interior == 0), its peers have
    # interior cells: all three must take the same (overlapped) exchange channel (comm.hip dist_apply)
    fractions = [0.0, 0.49, 0.51, 1.0] if method == "thin" else None
    deg = 2 if method == "p2" else 1
    S = distribute_solver(lambda: SolverIdealized(params, ion_list, degree_emi=deg, degree_knp=deg), mesh_tuple, ode_models, stim_params,
                          rank, world, 0, None, method="slab" if method in ("thin", "p2") else method, fractions=fractions)
    if method == "thin":
        assert (S.dev.n_interior == 0) == (rank == 1), (rank, S.dev.n_interior)
    S._unpack_solver_params(solver_parameters(3, 0)._replace(rtol_emi=1e-10, rtol_knp=1e-12))
S.save_fields = S.save_solver_stats = False
S.splitting_scheme = True
S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
t = Constant(0.0)
for k in range(steps):
    S.step_membrane_models(k)
    S.solve_for_time_step(k, t)
loc = S.local_mesh
n_own = loc.nc_owned
nc = loc.mesh.num_cells()
c = S.c.array().reshape(S.N_ions, nc, S.nd)[:, :n_own]
phi = S.phi.array().reshape(nc, S.nd)[:n_own]
np.savez(os.path.join(outdir, "rank%d.npz" % rank), cells=loc.cells_global[:n_own], c=c, phi=phi, emi_its=np.asarray(S.emi_niter),
         knp_its=np.asarray([max(n) for n in S.knp_niter]), dist0=int(getattr(S, "amg_dist0", 0)), emi_targets=np.asarray(S.emi_targets))
S.dev.close()
