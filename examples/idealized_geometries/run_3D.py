#!/usr/bin/python3
"""3D idealized geometry, 4 axons in ECS, HH membranes (reference: examples/idealized-geometries/run_3D.py)."""
import sys
from idealized_common import make_solver, solver_parameters, Constant

if __name__ == "__main__":
    resolution = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    Tstop = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0e-2
    # third argument "nosave": the same run without the per-step field output (the reference always writes it, run_3D.py:205-206) --
    # the compute-only end-to-end time of the example
    save = not (len(sys.argv) > 3 and sys.argv[3] == "nosave")
    S = make_solver(dim=3, resolution=resolution, verbose=save)
    t = Constant(0.0)
    S.solve_system_active(Tstop, t, solver_parameters(3, resolution), filename="results/data/3D/",
                          save_fields=save, save_solver_stats=True)
    if not save:
        import numpy as np
        print("steps %d  EMI iterations %.2f per step  KNP %.2f  (solve timers: EMI %.3f s, KNP %.3f s, assembly %.3f s, ODE %.3f s)"
              % (len(S.emi_niter), np.mean(S.emi_niter), np.mean([max(n) for n in S.knp_niter]), S.emi_solve_timer, S.knp_solve_timer,
                 S.emi_ass_timer + S.knp_ass_timer, S.ode_solve_timer))
        try:
            import psutil, time
            print("wall since process start: %.2f s" % (time.time() - psutil.Process().create_time()))
        except ImportError:
            pass
