"""Cell partition of the mesh over the GPUs of one node + one layer of facet-neighbour ghost cells.

DG couples cells only through shared facets (dS integrals, reference: src/knpemidg/solver.py:325-328,
586-594), so every rank needs exactly its owned cells plus their facet neighbours.  The reference gets
this from DOLFIN's ghost_mode + PETSc scatters (solver.py:16, 529, 789) and never tests it; here it is
explicit:

* owned cells   : contiguous chunks of the cells sorted by centroid x (x-slabs for the idealized
                  BoxMesh geometries: 2 peers per rank, every peer one xGMI hop);
* local numbering: owned cells first (global order kept), then ghosts grouped by owner rank, each group
                  in ascending global cell id -> a peer's ghosts are one contiguous receive range;
* send lists    : owned cells that a peer holds as ghosts, ascending global id (same order the peer
                  expects); computed redundantly on every rank from the global facet table, no
                  communication needed;
* membrane facets on a partition boundary are handled by BOTH ranks (phi_M, E_k and the ODE step are
  deterministic functions of the two adjacent cells, which both ranks hold), so no facet exchange exists.
"""
import numpy as np

from knpemidg.mesh import Mesh, MeshFunction


class Partition:
    def __init__(self, mesh, world, axis=0):
        self.mesh = mesh
        self.world = int(world)
        nc = mesh.num_cells()
        cm = mesh.cell_midpoints()[:, axis]
        order = np.argsort(cm, kind="stable")
        owner = np.empty(nc, dtype=np.int32)
        bounds = [(nc * r) // self.world for r in range(self.world + 1)]
        for r in range(self.world):
            owner[order[bounds[r]:bounds[r + 1]]] = r
        self.owner = owner
        fc = mesh.facet_cells
        it = fc[:, 1] >= 0
        c0, c1 = fc[it, 0], fc[it, 1]
        cut = owner[c0] != owner[c1]
        self._cut = (c0[cut], c1[cut])

    def ghosts_of(self, rank):
        """{owner q: ascending global ids of the cells rank needs from q}."""
        c0, c1 = self._cut
        o0, o1 = self.owner[c0], self.owner[c1]
        need = np.concatenate([c1[o0 == rank], c0[o1 == rank]])
        need = np.unique(need)
        own = self.owner[need]
        return {int(q): need[own == q] for q in np.unique(own)}

    def local(self, rank):
        return LocalMesh(self, rank)


class LocalMesh:
    """Sub-mesh of one rank: owned + ghost cells, local vertex numbering, halo tables."""

    def __init__(self, part, rank):
        mesh = part.mesh
        self.rank = rank
        self.owned = np.nonzero(part.owner == rank)[0]
        ghosts = part.ghosts_of(rank)
        self.peers = sorted(ghosts.keys())
        self.recv_offsets, self.recv_counts = [], []
        cells_g = [self.owned]
        off = len(self.owned)
        for q in self.peers:
            self.recv_offsets.append(off)
            self.recv_counts.append(len(ghosts[q]))
            cells_g.append(ghosts[q])
            off += len(ghosts[q])
        self.cells_global = np.concatenate(cells_g)
        self.nc_owned = len(self.owned)
        g2l = np.full(mesh.num_cells(), -1, dtype=np.int64)
        g2l[self.cells_global] = np.arange(len(self.cells_global))
        self.g2l = g2l
        # send lists: what each peer holds of mine as ghosts
        self.send_lists = []
        for q in self.peers:
            mine = part.ghosts_of(q).get(rank, np.zeros(0, dtype=np.int64))
            self.send_lists.append(g2l[mine].astype(np.int32))
        # local vertex numbering keeps the ascending order inside cells
        cv = mesh.cells[self.cells_global]
        self.verts_global = np.unique(cv)
        v2l = np.full(mesh.num_vertices(), -1, dtype=np.int64)
        v2l[self.verts_global] = np.arange(len(self.verts_global))
        self.mesh = Mesh(mesh.coords[self.verts_global], v2l[cv])
        # local facet -> global facet through (cell, local facet index), which the renumbering preserves
        lf = self.mesh.cell_facets
        gf = mesh.cell_facets[self.cells_global]
        self.facets_global = np.empty(self.mesh.num_facets(), dtype=np.int64)
        self.facets_global[lf.ravel()] = gf.ravel()

    def localize(self, subdomains, surfaces, membrane_tags):
        """Local tag arrays.  Membrane facets that touch no owned cell are demoted to tag 0 so that no
        ODE node is created for them on this rank."""
        st = np.asarray(subdomains.array() if hasattr(subdomains, "array") else subdomains)[self.cells_global]
        ft = np.asarray(surfaces.array() if hasattr(surfaces, "array") else surfaces)[self.facets_global].copy()
        fc = self.mesh.facet_cells
        touches_owned = (fc[:, 0] < self.nc_owned) | ((fc[:, 1] >= 0) & (fc[:, 1] < self.nc_owned))
        inactive = np.isin(ft, list(membrane_tags)) & ~touches_owned
        ft[inactive] = 0
        return MeshFunction(self.mesh, self.mesh.gdim, st), MeshFunction(self.mesh, self.mesh.gdim - 1, ft)

    def exchange_host(self, arr, dist):
        """Reference halo exchange on host arrays [n_local_cells, ...] through torch.distributed (gloo on the
        CPU: used by the tests to validate these tables; the product path exchanges on the GPU via RCCL)."""
        import torch
        reqs, bufs = [], []
        for q, sl, ro, rc in zip(self.peers, self.send_lists, self.recv_offsets, self.recv_counts):
            s = torch.from_numpy(np.ascontiguousarray(arr[sl]))
            r = torch.empty((rc,) + arr.shape[1:], dtype=s.dtype)
            reqs.append(dist.isend(s, q))
            reqs.append(dist.irecv(r, q))
            bufs.append((ro, rc, r, s))
        for rq in reqs:
            rq.wait()
        for ro, rc, r, _ in bufs:
            arr[ro:ro + rc] = r.numpy()
        return arr


def make_distributed_solver(dim, resolution, rank, world, local_rank, dist, n_axons=4, degree=1, dt=1.0e-4,
                            solver_cls=None, mesh_tuple=None):
    """The idealized-geometry solver on `world` GPUs: every rank builds the global mesh (cheap: numpy),
    keeps its slab + ghosts, creates its device context and joins the RCCL communicator."""
    import os
    import sys
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples",
                      "idealized_geometries")
    if ex not in sys.path:
        sys.path.insert(0, ex)
    from idealized_common import SolverIdealized, physical_setup
    from knpemidg.mesh import make_mesh_2D, make_mesh_3D
    from knpemidg.models import mm_hh, mm_hh_no_stim
    from knpemidg import _abi

    if mesh_tuple is None:
        mesh_tuple = make_mesh_3D(resolution, n_axons=n_axons) if dim == 3 else make_mesh_2D(resolution)
    mesh, subdomains, surfaces = mesh_tuple
    if dim == 3:
        ode_models = {1: mm_hh, 2: mm_hh_no_stim} if n_axons > 1 else {1: mm_hh}
    else:
        ode_models = {1: mm_hh}
    part = Partition(mesh, world)
    loc = part.local(rank)
    sub_l, surf_l = loc.localize(subdomains, surfaces, ode_models.keys())
    params, ion_list, stim_params = physical_setup(dt)
    cls = solver_cls or SolverIdealized
    S = cls(params, ion_list, degree_emi=degree, degree_knp=degree)
    S.verbose = False
    S.device_index = local_rank
    S.nc_owned = loc.nc_owned
    S.global_num_cells = mesh.num_cells()
    S.local_mesh = loc
    S.global_mesh_tuple = mesh_tuple
    S.setup_domain(loc.mesh, sub_l, surf_l)
    S.setup_parameters()
    S.setup_FEM_spaces()
    S.setup_membrane_model(stim_params, ode_models)
    # RCCL communicator: rank 0 creates the id, torch.distributed carries it to the others
    uid = [_abi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    S.dev.comm_init(rank, world, uid[0])
    S.dev.halo_tables(loc.peers, loc.send_lists, loc.recv_offsets, loc.recv_counts)
    return S
