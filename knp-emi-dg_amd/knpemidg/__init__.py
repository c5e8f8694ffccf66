"""knpemidg -- MI355X-native drop-in for the DG assemble-and-solve path of adajel/KNP-EMI-DG.

Mirrors the export list of the reference package (reference: src/knpemidg/__init__.py:1-17);
the dolfin-typed arguments are replaced by the array-backed stand-ins of `knpemidg.mesh`.
"""
from knpemidg.mesh import Mesh, MeshFunction, Constant, RectangleMesh, BoxMesh
from knpemidg.mesh import make_mesh_2D, make_mesh_3D, make_mesh_MMS

from knpemidg.membrane import MembraneModel, get_indices, is_dlt_scalar, get_values, set_values
from knpemidg.utils import (subdomain_marking_foo, interface_normal, plus, minus, pcws_constant_project,
                            CellCenterDistance)
from knpemidg.solver import Solver
from knpemidg.solver_emi import SolverEMI

__all__ = ["Solver", "SolverEMI", "MembraneModel", "subdomain_marking_foo", "interface_normal", "plus", "minus",
           "pcws_constant_project", "CellCenterDistance", "Mesh", "MeshFunction", "Constant", "RectangleMesh", "BoxMesh",
           "make_mesh_2D", "make_mesh_3D", "make_mesh_MMS"]
