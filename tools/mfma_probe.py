"""FP64 MFMA vs vector FMA on the DG-P2 facet-quadrature contraction (knp_probe_facet_contraction): prints the average kernel
time of both variants for the facet counts of the r=1 and r=2 meshes.  usage: mfma_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import knpemi_oracle as ko
from common import device_for
from knpemidg.mesh import make_mesh_2D
m, s, f = make_mesh_2D(0)
dev = device_for(ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,)))
rng = np.random.default_rng(0)
for cells in (124416, 995328):
    ncol = 4 * cells
    a = rng.uniform(0.5, 1.5, size=(ncol, 26))
    res = {}
    for variant, name in ((0, "vector FMA chain"), (1, "v_mfma_f64_16x16x4")):
        out, ms = dev.probe_facet_contraction(variant, a, reps=50)
        res[name] = (ms, out)
        flops_useful = ncol * 12 * (3 * 6 + 2 * 3 + 6 + 3) * 2
        print("%8d cells x 4 facets  %-20s %8.1f us   %6.2f TFLOP/s useful   %5.1f GB/s of input+output" %
              (cells, name, 1e3 * ms, flops_useful / ms / 1e9, ncol * 35 * 8 / ms / 1e6))
    d = np.abs(res["vector FMA chain"][1] - res["v_mfma_f64_16x16x4"][1]).max()
    print("          max difference between the variants: %.2e" % d)
dev.close()
