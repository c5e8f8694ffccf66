#!/usr/bin/python3
"""2D idealized single neuron in ECS, HH membrane (reference: examples/idealized-geometries/run_2D.py)."""
import sys
from idealized_common import make_solver, solver_parameters, Constant

if __name__ == "__main__":
    resolution = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    Tstop = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0e-2
    S = make_solver(dim=2, resolution=resolution, verbose=True)
    t = Constant(0.0)
    S.solve_system_active(Tstop, t, solver_parameters(2, resolution), filename="results/data/2D/",
                          save_fields=True, save_solver_stats=True)
