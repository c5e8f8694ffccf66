import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "idealized_geometries"))
from idealized_common import make_solver, solver_parameters, Constant, physical_setup  # noqa: F401,E402
