"""Mesh and mesh-function files of the reference's drivers, without dolfin:

* DOLFIN XML (`Mesh(mesh_path)`, `MeshFunction('size_t', mesh, path)`; reference: examples/idealized-geometries/run_3D.py:166-168,
  written by make_mesh_3D.py:107-111): `<dolfin><mesh celltype dim><vertices><vertex index x y z/>..<cells><tetrahedron index v0..v3/>`
  and `<dolfin><mesh_function><mesh_value_collection type dim size><value cell_index local_entity value/>`.
  Local entity i of a cell is the facet opposite its i-th vertex in ASCENDING vertex order (UFC), which is this build's facet
  convention too, so facet values map straight onto `Mesh.cell_facets`.
* XDMF + HDF5 reading lives in knpemidg/h5lite.py (read_xdmf_mesh); result files are written by Solver.init_h5_savefile.
"""
import re

import numpy as np

from knpemidg.mesh import Mesh, MeshFunction

_CELL = {"interval": 2, "triangle": 3, "tetrahedron": 4}


def read_dolfin_xml_mesh(path):
    """Mesh from a DOLFIN XML file (cells re-sorted to ascending vertex ids, as dolfin's ordering does)."""
    txt = open(path).read()
    m = re.search(r'<mesh\s+celltype="(\w+)"\s+dim="(\d+)"', txt)
    if not m:
        raise ValueError("%s: not a DOLFIN XML mesh" % path)
    nvc, dim = _CELL[m.group(1)], int(m.group(2))
    keys = ("x", "y", "z")[:dim]
    vpat = r'<vertex\s+index="(\d+)"\s+' + r'\s+'.join(r'%s="([^"]+)"' % k for k in keys)
    v = np.array(re.findall(vpat, txt), dtype=np.float64)
    coords = np.zeros((len(v), dim))
    coords[v[:, 0].astype(np.int64)] = v[:, 1:]
    cpat = r'<%s\s+index="(\d+)"\s+' % m.group(1) + r'\s+'.join(r'v%d="(\d+)"' % k for k in range(nvc))
    c = np.array(re.findall(cpat, txt), dtype=np.int64)
    cells = np.zeros((len(c), nvc), dtype=np.int64)
    cells[c[:, 0]] = np.sort(c[:, 1:], axis=1)
    n_v, n_c = re.search(r'<vertices\s+size="(\d+)"', txt), re.search(r'<cells\s+size="(\d+)"', txt)
    if (n_v and int(n_v.group(1)) != len(coords)) or (n_c and int(n_c.group(1)) != len(cells)):
        raise ValueError("%s: vertex / cell count does not match the size attributes" % path)
    return Mesh(coords, cells)


def read_dolfin_xml_meshfunction(mesh, path):
    """MeshFunction('size_t', mesh, path): cell (dim = gdim) or facet (dim = gdim - 1) values."""
    txt = open(path).read()
    m = re.search(r'<mesh_value_collection[^>]*dim="(\d+)"', txt)
    if not m:
        raise ValueError("%s: not a DOLFIN XML mesh function" % path)
    dim = int(m.group(1))
    vals = np.array(re.findall(r'<value\s+cell_index="(\d+)"\s+local_entity="(\d+)"\s+value="(\d+)"', txt), dtype=np.int64)
    f = MeshFunction(mesh, dim, 0)
    if dim == mesh.gdim:
        f.array()[vals[:, 0]] = vals[:, 2]
    elif dim == mesh.gdim - 1:
        f.array()[mesh.cell_facets[vals[:, 0], vals[:, 1]]] = vals[:, 2]
    else:
        raise NotImplementedError("mesh functions over entities of dimension %d" % dim)
    return f


def write_dolfin_xml_mesh(mesh, path):
    names = {v: k for k, v in _CELL.items()}
    nv = mesh.cells.shape[1]
    with open(path, "w") as fh:
        fh.write('<?xml version="1.0"?>\n<dolfin xmlns:dolfin="http://fenicsproject.org">\n')
        fh.write('  <mesh celltype="%s" dim="%d">\n    <vertices size="%d">\n' % (names[nv], mesh.gdim, mesh.num_vertices()))
        for i, x in enumerate(mesh.coords):
            fh.write('      <vertex index="%d" %s />\n' % (i, " ".join('%s="%.16e"' % (k, v) for k, v in zip("xyz", x))))
        fh.write('    </vertices>\n    <cells size="%d">\n' % mesh.num_cells())
        for i, c in enumerate(mesh.cells):
            fh.write('      <%s index="%d" %s />\n' % (names[nv], i, " ".join('v%d="%d"' % (k, v) for k, v in enumerate(c))))
        fh.write('    </cells>\n  </mesh>\n</dolfin>\n')


def write_dolfin_xml_meshfunction(mesh, f, path):
    dim = f.dim()
    a = np.asarray(f.array())
    with open(path, "w") as fh:
        fh.write('<?xml version="1.0"?>\n<dolfin xmlns:dolfin="http://fenicsproject.org">\n  <mesh_function>\n')
        if dim == mesh.gdim:
            fh.write('    <mesh_value_collection name="f" type="uint" dim="%d" size="%d">\n' % (dim, len(a)))
            for c, v in enumerate(a):
                fh.write('      <value cell_index="%d" local_entity="0" value="%d" />\n' % (c, v))
        else:
            cf = mesh.cell_facets
            fh.write('    <mesh_value_collection name="f" type="uint" dim="%d" size="%d">\n' % (dim, cf.size))
            for c in range(cf.shape[0]):
                for i in range(cf.shape[1]):
                    fh.write('      <value cell_index="%d" local_entity="%d" value="%d" />\n' % (c, i, a[cf[c, i]]))
        fh.write('    </mesh_value_collection>\n  </mesh_function>\n</dolfin>\n')
