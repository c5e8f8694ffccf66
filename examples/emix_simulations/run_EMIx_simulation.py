"""EMIx simulation (reference: examples/emix-simulations/run_EMIx_simulation.py): 121 617-tet reconstruction with
glial (tag 1) and neuronal (tag 2) membranes.  usage: python run_EMIx_simulation.py [--Tstop 1.0] [--out results/]"""
import argparse
import time

from emix_common import make_solver, solver_parameters, Constant

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--Tstop", type=float, default=1.0, help="end time in ms (dt = 0.1 ms)")
    ap.add_argument("--out", default=None, help="directory for .npz field snapshots")
    args = ap.parse_args()
    S = make_solver(verbose=False)
    print("EMIx mesh: %d tets, %d membrane facets, %d DoFs" % (S.mesh.num_cells(), S.dev.nmf if hasattr(S.dev, "nmf") else -1,
                                                              S.mesh.num_cells() * S.nd * (1 + S.N_ions)))
    t = Constant(0.0)
    t0 = time.perf_counter()
    S.solve_system_active(args.Tstop, t, solver_parameters(), filename=args.out, save_fields=args.out is not None)
    n = max(1, len(S.emi_niter))
    print("steps %d  %.1f ms/step  EMI its %s  KNP its %s" % (n, 1e3 * (time.perf_counter() - t0) / n, S.emi_niter,
                                                             [max(k) for k in S.knp_niter]))
