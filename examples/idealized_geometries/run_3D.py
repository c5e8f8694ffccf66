#!/usr/bin/python3
"""3D idealized geometry, 4 axons in ECS, HH membranes (reference: examples/idealized-geometries/run_3D.py)."""
import sys
from idealized_common import make_solver, solver_parameters, Constant

if __name__ == "__main__":
    resolution = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    Tstop = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0e-2
    S = make_solver(dim=3, resolution=resolution, verbose=True)
    t = Constant(0.0)
    S.solve_system_active(Tstop, t, solver_parameters(3, resolution), filename="results/data/3D/",
                          save_fields=True, save_solver_stats=True)
