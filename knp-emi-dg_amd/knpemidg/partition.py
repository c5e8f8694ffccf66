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


def rcb_owner(points, world):
    """Recursive coordinate bisection of `points` [n, d] into `world` parts of (almost) equal size: split the longest axis of
    the current box at the position that divides the cells in proportion to the ranks on either side.  Deterministic, so
    every rank computes the same partition without communication (no METIS in this environment)."""
    pts = np.asarray(points, dtype=np.float64)
    owner = np.empty(len(pts), dtype=np.int32)

    def split(idx, r0, nr):
        if nr == 1:
            owner[idx] = r0
            return
        p = pts[idx]
        axis = int(np.argmax(p.max(axis=0) - p.min(axis=0)))
        nl = nr // 2
        k = (len(idx) * nl) // nr
        order = idx[np.argsort(p[:, axis], kind="stable")]
        split(order[:k], r0, nl)
        split(order[k:], r0 + nl, nr - nl)
    split(np.arange(len(pts)), 0, int(world))
    return owner


class Partition:
    """method "slab": contiguous chunks of the cells sorted by centroid coordinate `axis` (idealized BoxMesh geometries: <= 2
    peers per rank); "rcb": recursive coordinate bisection (unstructured meshes, e.g. the EMIx reconstruction)."""

    def __init__(self, mesh, world, axis=0, method="slab", fractions=None):
        # fractions (slab only): world + 1 increasing numbers from 0 to 1, the share of the sorted cells below every cut (default:
        # equal parts); lets a test build a rank so thin that ALL its cells touch a cut
        self.mesh = mesh
        self.world = int(world)
        nc = mesh.num_cells()
        if method == "rcb":
            owner = rcb_owner(mesh.cell_midpoints(), self.world)
        elif method == "slab":
            cm = mesh.cell_midpoints()[:, axis]
            order = np.argsort(cm, kind="stable")
            owner = np.empty(nc, dtype=np.int32)
            bounds = [(nc * r) // self.world for r in range(self.world + 1)] if fractions is None else \
                [int(round(nc * float(f))) for f in fractions]
            assert len(bounds) == self.world + 1 and bounds[0] == 0 and bounds[-1] == nc and all(b1 > b0 for b0, b1 in zip(bounds, bounds[1:]))
            for r in range(self.world):
                owner[order[bounds[r]:bounds[r + 1]]] = r
        else:
            raise ValueError("partition method must be 'slab' or 'rcb'")
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
        self.part = part
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


def distribute_solver(solver_factory, mesh_tuple, ode_models, stim_params, rank, world, local_rank, dist, method="rcb", fractions=None):
    """Any `Solver` subclass on `world` GPUs of one node.  `solver_factory()` returns a fresh, un-set-up solver (its params and
    ion list inside); `mesh_tuple` = (mesh, subdomains, surfaces) of the GLOBAL mesh, which every rank holds (host memory only);
    `ode_models` = {membrane facet tag: ODE module}.  Every rank keeps its part + one ghost layer, creates its device context,
    joins the two RCCL communicators (reductions / halo exchanges) and receives its halo tables."""
    from knpemidg import _abi
    mesh, subdomains, surfaces = mesh_tuple
    part = Partition(mesh, world, method=method, fractions=fractions)
    loc = part.local(rank)
    sub_l, surf_l = loc.localize(subdomains, surfaces, ode_models.keys())
    S = solver_factory()
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
    import os
    shm = os.environ.get("KNP_COMM_SHM")
    if shm:
        # validation runs with several ranks on ONE GPU (RCCL refuses that): host-staged shared-memory communicator, no torch.distributed
        # (every rank must ask for the same segment layout: capacities from global quantities only)
        ncg_bound = 4 * (mesh.coords.shape[0] + 8 * mesh.num_cells() // 4)            # conforming P1 / P2 dofs, two columns
        most_sent = max(sum(len(sl) for sl in part.local(r).send_lists) for r in range(world))
        # outbox: the cell halo ([field <= 7][cell][nd]) or the interface exchange of the row-distributed conforming level
        # ([shared dof, peer][column <= 7]; a cut cell touches nd conforming dofs, each shared with at most a handful of peers)
        S.dev.comm_init_shm(rank, world, shm, max(ncg_bound, 1 << 16), max(most_sent, 1) * 7 * S.nd * 4)
    else:
        # RCCL communicators: rank 0 creates the ids, torch.distributed carries them to the others
        uid = [(_abi.comm_unique_id(), _abi.comm_unique_id()) if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        S.dev.comm_init(rank, world, uid[0][0], uid_halo=uid[0][1])
    S.dev.halo_tables(loc.peers, loc.send_lists, loc.recv_offsets, loc.recv_counts)
    return S


def make_distributed_solver(dim, resolution, rank, world, local_rank, dist, n_axons=4, degree=1, dt=1.0e-4,
                            solver_cls=None, mesh_tuple=None):
    """The idealized-geometry solver on `world` GPUs (x-slab partition of the BoxMesh: 2 peers per rank, each one xGMI hop)."""
    import os
    import sys
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples",
                      "idealized_geometries")
    if ex not in sys.path:
        sys.path.insert(0, ex)
    from idealized_common import SolverIdealized, physical_setup
    from knpemidg.mesh import make_mesh_2D, make_mesh_3D
    from knpemidg.models import mm_hh, mm_hh_no_stim

    if mesh_tuple is None:
        mesh_tuple = make_mesh_3D(resolution, n_axons=n_axons) if dim == 3 else make_mesh_2D(resolution)
    if dim == 3:
        ode_models = {1: mm_hh, 2: mm_hh_no_stim} if n_axons > 1 else {1: mm_hh}
    else:
        ode_models = {1: mm_hh}
    params, ion_list, stim_params = physical_setup(dt)
    cls = solver_cls or SolverIdealized
    return distribute_solver(lambda: cls(params, ion_list, degree_emi=degree, degree_knp=degree), mesh_tuple, ode_models,
                             stim_params, rank, world, local_rank, dist, method="slab")
