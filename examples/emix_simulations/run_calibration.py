"""ODE-only calibration run behind the initial state of the EMIx simulations (reference:
examples/emix-simulations/run_calibration.py:13-90 with mm_calibration.py): the extended membrane system -- neuronal Hodgkin-Huxley
membrane, glial Kir membrane and the three compartment concentrations -- is stepped with the membrane integrator until it is
stationary; the end state gives phi_M, the gating variables and the ECS / neuron / glia concentrations that
run_EMIx_simulation.py:76-84 and mm_hh.py / mm_glial.py start from.  Here the steps run on the device (csrc/ode.hip, model 6)
through the same `MembraneModel.step_lsoda` the PDE runs use.

usage: python run_calibration.py [n_steps]      (reference: 100000 steps of 0.1 ms)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(os.path.dirname(HERE)), "knp-emi-dg_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

from knpemidg import _abi                                                     # noqa: E402
from knpemidg.mesh import RectangleMesh, MeshFunction                         # noqa: E402
from knpemidg.functions import FacetSpace                                     # noqa: E402
from knpemidg.membrane import MembraneModel                                   # noqa: E402
from knpemidg.models import mm_calibration as ode                             # noqa: E402

NAMES = (("phi_M_n_init", "V_n"), ("phi_M_g_init", "V_g"), ("K_e_init", "K_e"), ("K_n_init", "K_n"), ("K_g_init", "K_g"),
         ("Na_e_init", "Na_e"), ("Na_n_init", "Na_n"), ("Na_g_init", "Na_g"), ("n_init", "n"), ("m_init", "m"), ("h_init", "h"))


def calibrate(n_steps=100000, dt=0.1, verbose=True):
    """Returns {name: steady-state value} after n_steps membrane steps of length dt (ms), g_syn_bar = 0 (run_calibration.py:20-21)."""
    mesh = RectangleMesh((0.0, 0.0), (1.0, 1.0), 2, 2)                         # df.UnitSquareMesh(2, 2), run_calibration.py:13
    facet_f = MeshFunction(mesh, 1, 0)
    V = FacetSpace(mesh)
    dev = _abi.Device(mesh, np.zeros(mesh.num_cells(), dtype=np.uint32), facet_f.array(), (), 3)
    membrane = MembraneModel(ode, facet_f=facet_f, tag=0, V=V)
    if not membrane.attach_device(dev):
        raise RuntimeError("the calibration system has no device implementation")
    stimulus = {'stim_amplitude': 0}
    for _ in range(n_steps):
        membrane.step_lsoda(dt=dt, stimulus=stimulus)
    states = membrane.states
    row = min(2, states.shape[0] - 1)                                         # the reference prints node 2 (all nodes are identical)
    out = {name: float(states[row, ode.state_indices(key)]) for name, key in NAMES}
    if verbose:
        for name, _ in NAMES:
            print(name, "=", out[name])
    dev.close()
    return out, states


if __name__ == "__main__":
    calibrate(int(sys.argv[1]) if len(sys.argv) > 1 else 100000)
