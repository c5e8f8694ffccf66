/* libknpemi_hip.so -- C ABI of the MI355X-native DG assemble-and-solve path of knpemidg.Solver.
 *
 * This is the drop-in seam that replaces, inside adajel/KNP-EMI-DG's `Solver`,
 *   - UFL form assembly      dolfin.assemble(a_emi|L_emi|B_emi|A_knp|L_knp)   src/knpemidg/solver.py:452-453,477-479,710,730-731
 *   - the PETSc matrix build  as_backend_type(...).mat()                      src/knpemidg/solver.py:458-460,482-484,711-712,734-735
 *   - the PETSc KSP solves    ksp.solve (cg / gmres + hypre)                  src/knpemidg/solver.py:509,755,771
 *   - the step-III projections pcws_constant_project / project               src/knpemidg/solver.py:808-845, utils.py:100-124
 * with matrix-free HIP kernels.  All functions are extern "C", take plain pointers and sizes,
 * return 0 on success and a negative status on failure (message: knp_last_error).  The caller
 * owns every host buffer; the context owns every device buffer.  One host thread per context,
 * one HIP stream per context; calls are synchronous with respect to the host unless noted.
 *
 * DoF layout (build-defined): scalar DG-p field  dof(c, j) = c*nd + j ;
 * species-major for the KNP unknown [k][c][j] ; facet fields one value per facet.
 * Cells hold ASCENDING vertex ids; local facet i is opposite local vertex i.
 */
#ifndef KNPEMI_HIP_H
#define KNPEMI_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct knp_ctx knp_ctx;

/* device-resident fields addressable through knp_upload / knp_download / knp_*_apply */
enum knp_field {
    KNP_F_PHI = 0,       /* potential phi                    [nc*nd]          Solver.phi          solver.py:165 */
    KNP_F_C = 1,         /* solved concentrations c          [n_sys][nc*nd]   Solver.c / c_prev_k solver.py:172-176 */
    KNP_F_C_PREV = 2,    /* c at previous time level         [n_sys][nc*nd]   Solver.c_prev_n     solver.py:174 */
    KNP_F_C_ELIM = 3,    /* eliminated ion                   [nc*nd]          ion_list[-1]['c']   solver.py:191 */
    KNP_F_PHI_M = 4,     /* membrane potential on facets     [nf]             phi_M_prev_PDE      solver.py:214 */
    KNP_F_I_CH = 5,      /* channel currents per ion         [n_ions][nf]     mem_models[..]['I_ch_k'] solver.py:251-259 */
    KNP_F_E = 6,         /* Nernst potentials per ion        [n_ions][nf]     ion['E']            solver.py:299-300 */
    KNP_F_KAPPA = 7,     /* kappa (derived)                  [nc*nd]          solver.py:306 */
    KNP_F_DNPHI = 8,     /* grad(phi).n per local facet      [nc*nd]          (derived, feeds solver.py:583,593) */
    KNP_F_B_EMI = 9,     /* assembled L_emi                  [nc*nd]          bb_emi              solver.py:478 */
    KNP_F_B_KNP = 10,    /* assembled L_knp                  [n_sys][nc*nd]   bb_knp              solver.py:731 */
    KNP_F_X = 11,        /* scratch vector                   [n_sys][nc*nd] */
    KNP_F_Y = 12,        /* scratch vector                   [n_sys][nc*nd] */
    KNP_F_FACET_TMP = 13,/* scratch facet fields             [KNP_FACET_TMP_SLOTS][nf]   results of pcws_constant_project */
    KNP_F_COUNT = 14
};

/* ---- context -------------------------------------------------------------------------------
 * Uploads the mesh and derives the per-(cell, local facet) neighbour / flag tables and the
 * membrane facet table.  Replaces Solver.setup_domain + interface_normal + subdomain_marking_foo
 * (solver.py:85-121, utils.py:44-85).
 *  cells        i32[nc][dim+1] vertex ids, local order = ascending ids of the caller's mesh (ids may then be
 *               relabelled for storage locality); cells [0,nc_owned) are owned, the rest are ghosts
 *  cell_tags    u32[nc]         subdomain tags (0 = ECS)
 *  facet_cells  i32[nf][2]      cells sharing each facet, -1 = none
 *  facet_local  i8 [nf][2]      local facet index within those cells
 *  facet_tags   u32[nf]         0 = ordinary interior facet (solver.py:60), membrane tags listed in membrane_tags
 */
int knp_ctx_create(knp_ctx** out, int device, int dim, int degree, int n_ions,
                   int64_t nv, int64_t nc, int64_t nc_owned, int64_t nf,
                   const double* coords, const int32_t* cells, const uint32_t* cell_tags,
                   const int32_t* facet_cells, const int8_t* facet_local, const uint32_t* facet_tags,
                   int n_membrane_tags, const uint32_t* membrane_tags);
void knp_ctx_destroy(knp_ctx* ctx);
const char* knp_last_error(knp_ctx* ctx);

/* Physical parameters (Solver.setup_parameters, solver.py:124-154; tau: solver.py:109-111).
 *  z[n_ions], D[n_ions][nc] (make_global, solver.py:1244-1258), rho[nc], fsrc[n_sys][nc] or NULL
 *  (ion['f_source'] on dx(0), solver.py:599), splitting: 1 = splitting scheme, 0 = original Robin
 *  data (solver.py:332-337, 614-622), 2 = manufactured-solution mode (knp_set_mms). */
int knp_set_params(knp_ctx* ctx, double C_M, double dt, double F, double R, double T, double C_phi,
                   double tau_emi, double tau_knp, const double* z, const double* D, const double* rho,
                   const double* fsrc, int splitting);

/* Optional geometry classes for (block-)structured meshes: cells with identical shape and neighbour-apex
 * positions share one 36-double record {vol, G upper triangle (10), per facet: L[4], sqrt(G_ii), 2/(h+h')}; the
 * operator applies then read no coordinates at all.  ncls = 0 switches back to the coordinate path. */
int knp_set_geometry_classes(knp_ctx* ctx, int ncls, const uint16_t* cls, const double* table);

/* Manufactured-solution mode (Solver(mms=...), solver.py:349-374, 632-657): splitting = 2 in knp_set_params selects
 * it.  C[n_sys][nc] are the DG0 coupling coefficients ion['C']; extra_emi[nc*nd] / extra_knp[n_sys][nc*nd] are the
 * solution-independent data terms (volume sources, Robin and flux-continuity data, Neumann data) integrated on the
 * host and added to L_emi / L_knp; the solution-dependent term -jump(phi) jump(C v) is evaluated on the device. */
int knp_set_mms(knp_ctx* ctx, const double* C, const double* extra_emi, const double* extra_knp);

/* Ion sources that are not constants (ion['f_source'] is any UFL coefficient in the reference, solver.py:599; e.g. the box- and
 * time-window Expression of examples/local-astrocyte-depolarization/run_tortuosity.py:180-200): src[n_sys][nc*nd] = the load
 * vector int f_k v dx(0) integrated by the caller, added to L_knp by knp_knp_rhs; null clears.  Constant sources travel as
 * fsrc[n_sys][nc] of knp_set_params. */
int knp_set_source(knp_ctx* ctx, const double* src);

/* DG-p path (degree 2): reference-basis tabulation of one integral class.  The operator applies are matrix-free and need
 * no tabulation (csrc/apply_p2.hip); the right-hand sides, facet averages and Nernst projections of step III (the device
 * analogue of assemble(L_emi / L_knp), solver.py:477-479, 730-731, and of pcws_constant_project, utils.py:100-124) are
 * integrated by numerical quadrature from these tables; the rules and the basis tabulated at their points come from the host
 * (knpemidg/dgtab.py).
 *  w[nq] weights summing to 1;  B[nloc][nq][nd] basis values;  dB[nloc][nq][nd][dim+1] derivatives with respect to
 *  the barycentric coordinates;  nloc = 1 for cell rules, dim+1 for facet rules (one tabulation per local facet, all
 *  with the same facet points: facet vertex m <-> cell vertex m + (m >= local facet index)).
 * Local dof order of a P2 cell: the dim+1 vertices, then the edge midpoints (a,b), a<b, lexicographic. */
enum knp_tab_slot {
    KNP_TAB_CELL_STIFF = 0,     /* dx of a_emi / a_knp                  degree max(2, 3p-2, 2p)  solver.py:318, 550-556 */
    KNP_TAB_CELL_RHS_EMI = 1,   /* dx of L_emi                          degree max(1, 2p-2)      solver.py:309-310 */
    KNP_TAB_CELL_MASS = 2,      /* dx of L_knp                          degree 2p                solver.py:597-599 */
    KNP_TAB_FACET_EMI = 3,      /* dS(0) of a_emi                       degree 3p                solver.py:321-328 */
    KNP_TAB_FACET_MEM = 4,      /* dS(membrane) of a_emi                degree 2p                solver.py:344 */
    KNP_TAB_FACET_KNP = 5,      /* dS(0) of a_knp                       degree max(2, 3p-1)      solver.py:586-594 */
    KNP_TAB_FACET_RHS_EMI = 6,  /* dS(0) of L_emi                       degree max(1, 2p-1)      solver.py:330 */
    KNP_TAB_FACET_MEM_LIN = 7,  /* dS(membrane) of L_emi                degree max(1, 2p-1)      solver.py:332-344 */
    KNP_TAB_FACET_MEM_KNP = 8,  /* dS(membrane) of L_knp                degree 5p                solver.py:603-629 */
    KNP_TAB_FACET_AVG = 9,      /* facet averages (phi_M, traces)       degree max(1, p)         utils.py:100-124 */
    KNP_TAB_FACET_NERNST = 10   /* ln(c_e/c_i)                          degree 2p+2              solver.py:827-828 */
};
int knp_set_tabulation(knp_ctx* ctx, int slot, int nloc, int nq, const double* w, const double* B, const double* dB);

int64_t knp_field_size(knp_ctx* ctx, int field);
int knp_upload(knp_ctx* ctx, int field, const double* src, int64_t offset, int64_t count);
int knp_download(knp_ctx* ctx, int field, double* dst, int64_t offset, int64_t count);
int knp_copy_field(knp_ctx* ctx, int dst_field, int src_field);

/* Inspection of the connectivity tables the library derives in knp_ctx_create from the raw cell / facet / tag arrays -- the
 * device counterpart of interface_normal / plus / minus (utils.py:61-98) and of the dS / membrane facet classification by tag
 * (solver.py:113-121).  `north_star` asks for bit-exact DoF / connectivity indexing: the parity tests read these tables back and
 * compare them entry for entry with the oracle's independently derived ones (tests/test_gpu_tables.py).  All tables are in DEVICE
 * cell order (the order of the `cells` argument of knp_ctx_create); facet ids are the caller's.
 *  knp_debug_table_size: number of BYTES of table `which`, < 0 if unknown
 *  knp_debug_table     : copies the table into out[nbytes]; nbytes must equal knp_debug_table_size */
enum knp_debug_table_id {
    KNP_DT_CELLS = 0,   /* int32  [nc][dim+1]  cell -> vertex (storage ids)                                                   */
    KNP_DT_NBR = 1,     /* int32  [nc][dim+1]  cell behind local facet i (opposite vertex i), -1 on the boundary              */
    KNP_DT_FLAG = 2,    /* uint32 [nc]         dim+1 flag bytes: bits 0-1 local facet index in the neighbour, bits 2-3 kind
                                               (0 SIPG, 1 membrane, 2 exterior, 3 inactive), bit 4 this cell is the `plus` side */
    KNP_DT_CFACET = 3,  /* int32  [nc][dim+1]  facet id behind local facet i                                                  */
    KNP_DT_MF = 4,      /* int32  [nmf][6]     membrane facets: plus cell, minus cell, their local facet indices, facet id,
                                               1 if a side is an owned cell                                                     */
    KNP_DT_HB_SRC = 5,  /* int32  [nblk][hs]   halo- / ring-staged applies: per 256-cell block, 4 * cell + local facet of every
                                               coupled (SIPG or membrane) neighbour outside the block, in (cell, facet) order, -1 padded (0 bytes if unused) */
    KNP_DT_HB_LOC = 6,  /* uint16 [nc_owned][4] LDS entry of the neighbour behind facet i: < 256 in-block, else 256 + list position */
    KNP_DT_META = 7     /* int64  [8]          nc, nc_owned, nf, nmf, hb_stride, hb_long0, n_interior, dim                     */
};
int64_t knp_debug_table_size(knp_ctx* ctx, int which);
int knp_debug_table(knp_ctx* ctx, int which, void* out, int64_t nbytes);

/* ---- per-step coefficient updates ----------------------------------------------------------- */
int knp_update_kappa(knp_ctx* ctx);        /* KAPPA <- C, C_ELIM                (solver.py:303-306) */
int knp_update_dnphi(knp_ctx* ctx);        /* DNPHI <- PHI                      (feeds solver.py:583,593) */

/* ---- operator applies (matrix-free MatMult) --------------------------------------------------
 * fy = A_emi(KAPPA) fx  on scalar fields;   fy = A_knp(DNPHI) fx  on [n_sys] species-major fields. */
int knp_emi_apply(knp_ctx* ctx, int fx, int fy);
int knp_knp_apply(knp_ctx* ctx, int fx, int fy);

/* ---- right-hand sides ------------------------------------------------------------------------ */
int knp_emi_rhs(knp_ctx* ctx);             /* B_EMI <- L_emi(C, C_ELIM, PHI_M, I_CH)          solver.py:478 */
int knp_knp_rhs(knp_ctx* ctx);             /* B_KNP <- L_knp(C, C_PREV, C_ELIM, PHI, PHI_M, I_CH) solver.py:731 */

/* ---- solves (KSP.solve) ----------------------------------------------------------------------
 * EMI: PCG, cell-block-Jacobi (+ auxiliary-space AMG when uploaded); initial guess = PHI (ksp_initial_guess_nonzero).
 *      Without a residual target (knp_emi_residual_target(ctx, 0), the state after knp_ctx_create): PETSc's default test on the
 *      preconditioned norm, ||M^-1 r|| <= max(rtol ||M^-1 b||, atol) (solver.py:425-444); res = {||M^-1 r0||, ||M^-1 r||, ||M^-1 b||}.
 *      With a residual target r_abs > 0 (what knpemidg.Solver uses; csrc/krylov.hip: cg_converged) the solve ends when BOTH
 *        (i)  ||(b - A phi) / vol||_8 <= r_abs      -- the TRUE residual in the order-8 norm of its density (sum_K (|r_K| / vol_K)^8)^(1/8),
 *             a sum-type stand-in for the max norm; the caller derives r_abs from the accuracy wanted in the concentrations, and
 *        (ii) ||phi - phi_k||_A <= rtol ||phi||_A   -- the energy-norm error of the iterate, estimated from the CG coefficients through the
 *             Hestenes-Stiefel identity ||x - x_k||_A^2 = sum_{j >= k} alpha_j (r_j . z_j), which holds for ANY SPD preconditioner
 *      hold, at the latest when ||M^-1 r|| <= 1e-11 ||M^-1 b|| (targets below what fp64 reaches must not loop forever); atol is not used;
 *      res = {||r0 / vol||_8, ||r / vol||_8, estimated ||phi - phi_k||_A / ||phi||_A}.  Neither (i) nor (ii) depends on the preconditioner.
 * KNP: per-species BiCGStab (or GMRES, knp_set_knp_krylov), same preconditioner family, TRUE residual: converged when
 *      ||r / vol||_8 <= max(20 rtol ||b / vol||_8, atol) -- order-8 norms of the residual and load DENSITIES: the concentrations are asked
 *      for in the max norm, and their max-norm error was measured at 0.03-0.055 of that ratio on uniform AND on sliver-ridden meshes
 *      (csrc/krylov.hip, profiles/r03_knp_norms_*.txt), i.e. the test asks for an estimated relative max-norm error of about rtol; the
 *      factor 20 (KNP_D8_FACTOR) applies to whatever rtol is passed (callers that mean the residual itself, e.g. a direct-solve
 *      emulation, divide it out); at least min_it iterations (ksp_min_it, solver.py:686); initial guess = C.
 *      res per system = {||r0 / vol||_8, ||r / vol||_8, ||b / vol||_8} (KNP_KNP_NORM2=1: the cell-volume-weighted 2-norms instead).
 * niter: iterations (EMI: 1 int, KNP: n_sys ints).  Returns -3 if not converged within maxit (ksp_error_if_not_converged, solver.py:428).
 * knp_emi_residual_target: r_abs > 0 arms the error-controlled stop above for the following EMI solves; 0 restores PETSc's test.
 *      Every rank of a partitioned run must pass the same number (knp_allreduce_sum). */
int knp_emi_residual_target(knp_ctx* ctx, double r_abs);
/* knp_knp_early_stop: factor in [0, 1); 0 (the state after knp_ctx_create) = a BiCGStab solve never stops before `min_it` iterations.  With
 * factor > 0 it also stops as soon as its residual is `factor` times UNDER the tolerance -- the floor exists to keep the per-step errors of a
 * quiet phase (the extrapolated initial guess already passes the test) far below the tolerance, which such a residual does by itself.
 * (GMRES keeps the plain floor.)  knpemidg.Solver sets 0.01 (`solver_params.knp_early_stop`). */
int knp_knp_early_stop(knp_ctx* ctx, double factor);
/* knp_knp_load_measure: out[k] = sum over this context's owned cells of (|b_K| / vol_K)^8 of the KNP right-hand side of species k (field B_KNP
 * as knp_knp_rhs left it; with KNP_KNP_NORM2=1: |b_K|^2 / vol_K) -- the size of the load the residual target above is scaled with
 * (knpemidg/solver.py: _knp_load_norm sums over the ranks and takes the root).  Replaces a host pass over the downloaded field. */
int knp_knp_load_measure(knp_ctx* ctx, double* out);
int knp_emi_solve(knp_ctx* ctx, double rtol, double atol, int maxit, int check_every, int* niter, double* res);
int knp_knp_solve(knp_ctx* ctx, double rtol, double atol, int maxit, int min_it, int check_every, int* niter, double* res);
/* Krylov method of knp_knp_solve: 0 = BiCGStab (default: 4 operator applies and no stored basis per iteration), 1 = right-preconditioned
 * restarted GMRES(restart), restart in 2..30 -- the reference's KSP type and restart length (ksp_type gmres, ksp_gmres_restart 30,
 * src/knpemidg/solver.py:684-701); same preconditioner and the same stopping test on the true residual either way.  niter of a GMRES
 * solve counts Arnoldi steps (= preconditioner applications). */
int knp_set_knp_krylov(knp_ctx* ctx, int method, int restart);
/* DG-level smoother of the EMI preconditioner (stands in for pc_type hypre on BB_emi, solver.py:433, 505): 1 = two-step Chebyshev
 * block-Jacobi (one more operator apply per PCG iteration, fewer iterations), 0 = plain cell-block-Jacobi, -1 = default (1 for DG-P1).
 * Every rank of a partitioned run must pass the same value (the preconditioner has to be one symmetric operator). */
int knp_set_emi_dg_smoother(knp_ctx* ctx, int chebyshev);

/* ---- auxiliary-space AMG preconditioner (stands in for pc_type hypre, solver.py:433, 688) ---------------
 * M^-1 = cell-block-Jacobi + P Ac^+ P^T with Ac the conforming (membrane-broken) P1 operator; the hierarchy is
 * built on the host (knpemidg/amg.py) and uploaded level by level.  which: 0 = EMI, 1 + k = KNP species k.
 *  knp_amg_begin : DG -> conforming dof map [nc*nd] and its inverse as a CSR list (conforming dof -> owned DG dofs, ascending;
 *                  cg_ptr = cg_idx = NULL: derived here)
 *  knp_amg_level : one level (A csr, inverse diagonal, spectral radius of D^-1 A, Chebyshev degree / lower
 *                  fraction; P [n x ncoarse] and R = P^T as CSR; ncoarse = 0 on the last level)
 *  knp_amg_finish: dense pseudo-inverse [n*n] of the last level; arms the preconditioner
 *  knp_amg_clear : back to plain block-Jacobi */
int knp_amg_begin(knp_ctx* ctx, int which, int64_t ncg, const int32_t* dg2cg, const int32_t* cg_ptr, const int32_t* cg_idx);
/* Optional, between knp_amg_begin and the first knp_amg_level: the hierarchy of slot `which` (>= 1) preconditions the
 * first ncol KNP species together (their level vectors become [ncol][n] and one chain of kernels carries all columns);
 * slots of the other species then stay empty.  Used when the species' diffusion coefficients are close. */
int knp_amg_columns(knp_ctx* ctx, int which, int ncol);
int knp_amg_level(knp_ctx* ctx, int which, int64_t n, const int32_t* rowptrA, const int32_t* colA, const double* valA,
                  const double* dinv, double rho, int cheb_degree, double cheb_lower, int64_t ncoarse,
                  const int32_t* rowptrP, const int32_t* colP, const double* valP,
                  const int32_t* rowptrR, const int32_t* colR, const double* valR);
int knp_amg_finish(knp_ctx* ctx, int which, int64_t n, const double* pinv);
int knp_amg_finish_f32(knp_ctx* ctx, int which, int64_t n, const float* pinv);   /* the same, inverse already in fp32 */
int knp_amg_clear(knp_ctx* ctx, int which);
/* Partitioned runs: ROW-DISTRIBUTED finest conforming level (the reference's BoomerAMG is row-distributed on every level under MPI,
 * src/knpemidg/solver.py:433, 688, with PETSc's VecScatter behind MatMult, :529, :789).  Each rank holds the conforming dofs of its own
 * cells (+ of the membrane facets assigned to it) in local numbering and the level-0 matrix SUB-ASSEMBLED from those cells / facets, so
 * that a product is a per-rank partial sum; partial sums at dofs shared by several ranks are exchanged point to point
 * (2 messages per peer) and added in rank order.  Levels >= 1 stay replicated behind ONE all-reduce of the level-1 residual per V-cycle.
 *  knp_amg_interface: the shared-dof tables (every rank of the communicator calls it, once per conforming space): peers[p] shares
 *                     counts[p] dofs; idx = their local numbers grouped by peer, ascending global dof within a peer; uvtx / aptr / asrc:
 *                     per distinct shared dof the message positions of its other owners in ascending rank order, -1 = own value.
 *  knp_amg_dist0    : between knp_amg_begin and the first knp_amg_level of a hierarchy uploaded in that local form. */
int knp_amg_interface(knp_ctx* ctx, int64_t n_local, int npeers, const int32_t* peers, const int64_t* counts, const int32_t* idx,
                      int64_t nuniq, const int32_t* uvtx, const int32_t* aptr, const int32_t* asrc);
int knp_amg_dist0(knp_ctx* ctx, int which);

/* ---- step III (solver.py:808-845) -------------------------------------------------------------
 * C_PREV <- C ; PHI_M <- facet-avg(phi_i - phi_e) ; C_ELIM <- -(sum z_k c_k + rho)/z_N ; E <- Nernst. */
int knp_step_updates(knp_ctx* ctx);
int knp_nernst(knp_ctx* ctx);              /* E only, from the current C / C_ELIM (solver.py:299-300) */
/* Picard variant (solve_for_time_step_picard, solver.py:850-927): per Picard level C_ELIM and E from the current C;
 * knp_max_abs_diff = inf-norm of the difference of two nodal fields (the eps of solver.py:879-880). */
int knp_picard_updates(knp_ctx* ctx);
int knp_max_abs_diff(knp_ctx* ctx, int field_a, int field_b, double* out);
/* FACET_TMP[slot] <- facet average of the plus (side 0, ECS-like) or minus (side 1) trace of a nodal field;
 * species indexes into [n_sys] fields (update_ode hook, examples/idealized-geometries/run_3D.py:39-51).  Several slots, so
 * that consecutive projections (K_e, Na_i, ...) do not overwrite each other before they are consumed. */
#define KNP_FACET_TMP_SLOTS 4
int knp_facet_trace(knp_ctx* ctx, int field, int species, int side, int slot);

/* ---- membrane ODEs: batched device integrator replacing the per-facet LSODA loop of
 * MembraneModel.step_lsoda (membrane.py:84-119).  model: 1 = Hodgkin-Huxley + stimulus (examples/idealized-geometries/mm_hh.py),
 * 2 = without (mm_hh_no_stim.py), 3 = EMIx neuron (examples/emix-simulations/mm_hh.py), 4 = EMIx glia (mm_glial.py), 5 = passive
 * leak (examples/rat-neuron/mm_leak.py), 6 = the ODE-only EMIx calibration system (examples/emix-simulations/mm_calibration.py).  Tables are [n][ns] states and [n][np] parameters in the reference's column layout.
 *  knp_ode_create  : returns a handle >= 0; facets[n] = facet id of every ODE node
 *  knp_ode_table   : what 0 = states, 1 = parameters; upload != 0 copies host -> device, else device -> host
 *  knp_ode_exchange: table column <- facet field (to_facet = 0, set_state/set_parameter) or facet field <- table
 *                    column (to_facet = 1, get_state/get_parameter); offset selects the row of [n_ions][nf] fields
 *  knp_ode_exchange_multi: n such copies in ONE launch (what[k], col[k], field[k], offset[k]); a membrane step moves V, E_k,
 *                    K_e, Na_i in and V, I_ch_k out (solver.py:1086-1112)
 *  knp_ode_set_stimulus: parameter columns cols[k] take values[k] on the rows with mask[row] != 0 at the start of EVERY
 *                    knp_ode_step (the reference re-imposes the stimulus every step, membrane.py:98-104); 0 entries clears
 *  knp_ode_step    : adaptive Dormand-Prince 5(4) from t0 to t0+dt per node (rtol 1e-8, atol 0: membrane.py:112).
 *                    Asynchronous: a node that fails (`assert success`, membrane.py:113) raises a device flag that the next
 *                    knp_emi_solve / knp_knp_solve status poll, knp_ode_table or knp_sync reports as -4 */
int knp_ode_create(knp_ctx* ctx, int model, int64_t n, const int32_t* facets, int ns, int np, const double* states,
                   const double* params);
int knp_ode_table(knp_ctx* ctx, int handle, int what, int upload, double* host);
int knp_ode_exchange(knp_ctx* ctx, int handle, int what, int col, int field, int64_t offset, int to_facet);
int knp_ode_exchange_multi(knp_ctx* ctx, int handle, int n, const int32_t* what, const int32_t* col, const int32_t* field,
                           const int64_t* offset, int to_facet);
int knp_ode_set_stimulus(knp_ctx* ctx, int handle, int n_entries, const int32_t* cols, const double* values, const uint8_t* mask);
int knp_ode_step(knp_ctx* ctx, int handle, double t0, double dt, double rtol, double atol);

/* ---- host-side setup kernels (no device work): threaded sparse products of the preconditioner setup (knpemidg/amg.py), the
 * counterpart of the BoomerAMG setup PETSc runs inside KSPSetUp (solver.py:433, 505, 688, 767).  CSR, int32 indices, fp64 values.
 *  knp_host_spgemm: C = A B, A [n x k], B [k x m]; Cp[n+1] supplied by the caller, *Cj / *Cx allocated here (knp_host_free);
 *                   sorted columns; nthreads <= 0 = all hardware threads; -3 if C exceeds 2^31 - 1 entries
 *  knp_host_spmv  : y = A x */
int knp_host_spgemm(int64_t n, int64_t m, const int32_t* Ap, const int32_t* Aj, const double* Ax, const int32_t* Bp, const int32_t* Bj,
                    const double* Bx, int32_t* Cp, int32_t** Cj, double** Cx, int nthreads);
int knp_host_spmv(int64_t n, const int32_t* Ap, const int32_t* Aj, const double* Ax, const double* x, double* y, int nthreads);
void knp_host_free(void* p);
/* Setup kernels of the conforming auxiliary operators (knpemidg/amg.py; the reference assembles BB_emi / AA_knp with dolfin.assemble at
 * every solve, src/knpemidg/solver.py:477-479, 730-731, and hands them to BoomerAMG):
 *  knp_host_cell_gram   : vol[nc], G[nc][(dim+1)^2] = grad lambda_a . grad lambda_b of every simplex cell from vertex coordinates
 *  knp_host_segment_sum : out[p] = sum_{k in [starts[p], starts[p+1])} src[order[k]] -- per-cell blocks summed into the values of a CSR
 *                         matrix whose pattern (sort order of the entry keys) is cached */
int knp_host_cell_gram(int64_t nc, int dim, const double* coords, const int32_t* cells, double* vol, double* G, int nthreads);
int knp_host_segment_sum(int64_t nseg, const int64_t* starts, const int64_t* order, const double* src, double* out, int nthreads);
/*  knp_host_block_pattern: the cached pattern itself -- entry (c, a, b) of blocks [nc][nd][nd] lands in (dof[c][a], dof[c][b]) of an
 *                         n x n matrix; order[nc nd nd], starts[<= nc nd nd + 1], cols[<= nc nd nd], indptr[n + 1] as a stable sort of the
 *                         entry keys would give them; *nseg = number of distinct (row, column) pairs */
/* Mesh tables on the host (knpemidg/mesh.py, knpemidg/_abi.py; the reference gets them from DOLFIN's mesh topology, solver.py:85-121):
 *  knp_host_build_facets     : facet numbering by first appearance in (cell, local facet) order + the facet -> cells / local index tables
 *  knp_host_geometry_classes : classes of cells with the same shape and neighbour configuration (structured meshes) */
int64_t knp_host_build_facets(int64_t nc, int nv, const int32_t* cells, int32_t* cell_facets, int32_t* facets, int32_t* facet_cells,
                              int8_t* facet_local);
int64_t knp_host_geometry_classes(int64_t nc, const double* coords, const int32_t* cells, const int32_t* nbr, const int8_t* nbj, double quantum,
                                  int64_t max_classes, int32_t* cls, int64_t* first, int nthreads);
int knp_host_block_pattern(int64_t nc, int nd, int64_t n, const int32_t* dof, int64_t* order, int64_t* starts, int32_t* cols, int32_t* indptr,
                           int64_t* nseg, int nthreads);
/*  knp_host_sym_to_f32: the dense coarsest-level inverse after LAPACK potri -- `a` [n x n] row-major with valid numbers in its upper triangle
 *  -> the full symmetric matrix rounded to fp32 (what knp_amg_finish_f32 uploads) and its largest magnitude (NaN if an entry is not finite) */
int knp_host_sym_to_f32(int64_t n, const double* a, float* out, double* maxabs, int nthreads);
/*  knp_host_mis2_aggregate: aggregates from a distance-2 maximal independent set of a strength graph (CSR pattern, symmetric, no diagonal),
 *  Luby rounds on the priorities key[n]; agg[n] receives the aggregate of every node, the return value is their number (< 0: bad arguments) */
int64_t knp_host_mis2_aggregate(int64_t n, const int32_t* indptr, const int32_t* indices, const double* key, int64_t* agg, int nthreads);
/*  knp_host_morton_order: stable argsort of points (or, with conn, of the midpoints of the rows of conn) along a Morton curve in units of
 *  `scale` per axis -- the device numbering of cells and vertices; knp_host_cell_extent_median: that scale, the median cell extent per axis */
int knp_host_morton_order(int64_t n, int d, const double* pts, const int32_t* conn, int nv, const double* scale, int64_t* order, int nthreads);
int knp_host_cell_extent_median(int64_t nc, int nv, int d, const double* coords, const int32_t* cells, double* out);
/*  knp_host_cell_neighbours: nb[nc][nv] = cell behind local facet i (-1: none), nj[nc][nv] = that cell's local index of the facet */
int knp_host_cell_neighbours(int64_t nc, int nv, const int32_t* cell_facets, const int32_t* facet_cells, const int8_t* facet_local, int32_t* nb,
                             int8_t* nj, int nthreads);
/*  knp_host_box_marks: out[i] = 1 where the midpoint of row i of conn lies in the box [a, b] (mode 0) / on its surface within eps (mode 1) */
int knp_host_box_marks(int64_t n, int d, const double* coords, const int32_t* conn, int nv, const double* a, const double* b, double eps, int mode,
                       uint8_t* out, int nthreads);
/*  Smoothed-aggregation passes (knpemidg/amg.py: build_hierarchy), row-parallel, the numbers of the numpy / scipy lines they replace:
 *  knp_host_strength            : pattern of the strong off-diagonal couplings |a_ij| >= theta sqrt(|a_ii a_jj|)           (*Sj: knp_host_free)
 *  knp_host_smooth_prolongator  : C = P - diag(v) A P, one damped-Jacobi step on a prolongator, exact zeros dropped         (*Cj, *Cx)
 *  knp_host_truncate_prolongator: rows cut below trunc * (row maximum) and rescaled to interpolate Bc to the same values    (*Tj, *Tx) */
int knp_host_strength(int64_t n, const int32_t* Ap, const int32_t* Aj, const double* Ax, const double* diag, double theta, int32_t* Sp, int32_t** Sj,
                      int nthreads);
int knp_host_smooth_prolongator(int64_t n, int64_t m, const int32_t* Ap, const int32_t* Aj, const double* Ax, const double* v, const int32_t* Pp,
                                const int32_t* Pj, const double* Px, int32_t* Cp, int32_t** Cj, double** Cx, int nthreads);
int knp_host_truncate_prolongator(int64_t n, const int32_t* Pp, const int32_t* Pj, const double* Px, double trunc, const double* Bc, int32_t* Tp,
                                  int32_t** Tj, double** Tx, int nthreads);

/* ---- timing / sync ------------------------------------------------------------------------------ */
int knp_sync(knp_ctx* ctx);
int knp_timer_begin(knp_ctx* ctx);         /* records a HIP event on the context's stream */
int knp_timer_end(knp_ctx* ctx, float* ms);/* records, synchronises, returns elapsed ms */
/* Launches `reps` back-to-back applies (which: 0 = EMI, 1 = KNP) on X -> Y and returns the average
 * kernel duration measured with HIP events on the stream the kernel runs on.  Three input / output vector pairs are used in
 * rotation so that the launches are not served from the Infinity Cache. */
int knp_bench_apply(knp_ctx* ctx, int which, int reps, float* avg_ms);
/* In-solver timing of the same kernels: while enabled, every operator apply launched by the solves is bracketed by a HIP event
 * pair on the context's stream; knp_apply_timing_read synchronises, returns the average duration and the number of launches
 * since the last read (which: 0 = EMI, 1 = KNP) and resets the tally. */
int knp_apply_timing(knp_ctx* ctx, int enable);
int knp_apply_timing_read(knp_ctx* ctx, int which, float* avg_ms, int* count);

/* Which kernel knp_emi_apply (which = 0) / knp_knp_apply (which = 1) currently dispatches to, so that a measurement can name it:
 * 0 coordinate path (any mesh), 1 geometry classes + LDS staging (structured 3D P1), 2 halo-staged persistent kernel, 6 the same
 * with D read through the material table (knp_set_params found <= 16 distinct coefficient tuples), 3 (EMI) / 7 (KNP) ring-staged
 * kernels (loader wave + LDS-DMA ring + consumer waves, csrc/apply_ring.hip; the default on structured 3D P1 meshes), 8 matrix-free P2,
 * 9 assembled P2 blocks (KNP_P2_ASSEMBLED=1).  Negative on bad arguments.  No reference counterpart (measurement only). */
int knp_apply_variant(knp_ctx* ctx, int which);

/* FP64 MFMA probe of the DG-P2 path's dense facet-quadrature contraction (csrc/apply_p2.hip): variant 0 = per-thread FMA chain
 * (what the product kernels use), 1 = v_mfma_f64_16x16x4_f64 tiles.  in[ncol][26] = {jump(u)[6], kappa[6], kappa'[6], dn u[3],
 * dn u'[3], area, penalty}, out[ncol][9] = {r[6], T[3]}; returns the average kernel time over `reps` launches.  Diagnostic
 * (no counterpart in the reference): it documents why the product path keeps the vector pipe. */
int knp_probe_facet_contraction(knp_ctx* ctx, int variant, int64_t ncol, int reps, const double* in, double* out, float* avg_ms);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) ------------------------------------------------
 * Replaces DOLFIN ghosting + PETSc VecGhost/MatMult scatters + KSP reductions (solver.py:16,529,789). */
int knp_comm_unique_id(char* out128);
int knp_comm_init(knp_ctx* ctx, int rank, int nranks, const char* id128);
/* Optional second communicator (its own unique id) + stream for the halo exchanges, so that the exchange of an apply's input
 * overlaps the launch over the interior cells; knp_set_interior declares how many leading owned cells have no ghost neighbour
 * (checked).  Without these two calls every exchange runs on the context's stream in front of the apply. */
int knp_comm_init_halo(knp_ctx* ctx, const char* id128);
/* Instead of knp_comm_init: a host-staged communicator through the POSIX shared-memory segment `name` for ranks that are processes
 * of one node and may share one GPU (RCCL refuses that).  Same semantics -- summed / maximised reductions in rank order, peer halo
 * exchange -- with every transfer staged through the host: a validation path (the partitioned solver with 2-4 ranks on a one-GPU
 * box, tests/test_gpu_multirank.py), never the measured one.  red_doubles / out_doubles: capacity of a rank's reduction slot and
 * halo outbox.  Call knp_halo_tables afterwards (it publishes where each peer finds its message and ends in a barrier). */
int knp_comm_init_shm(knp_ctx* ctx, int rank, int nranks, const char* name, int64_t red_doubles, int64_t out_doubles);
int knp_set_interior(knp_ctx* ctx, int64_t n_interior);
/* send_cells: owned cell ids whose DoFs peer p needs, grouped by peer; ghosts of peer p occupy
 * cells [recv_offsets[p], recv_offsets[p]+recv_counts[p]). */
int knp_halo_tables(knp_ctx* ctx, int npeers, const int32_t* peers, const int64_t* send_counts,
                    const int32_t* send_cells, const int64_t* recv_offsets, const int64_t* recv_counts);
int knp_halo_exchange(knp_ctx* ctx, int field);
/* Sum of n <= 56 host scalars over the ranks of a partitioned run (one ncclAllReduce on the solver's stream; every rank receives the
 * same bits); leaves the values untouched without a communicator.  The host side uses it where a quantity that steers the solve
 * must be the SAME on all ranks: the step-0 residual target of the EMI solve (the reference gets this from PETSc's VecNorm over
 * the MPI communicator, src/knpemidg/solver.py:505-509) and the timings of the smoother selection (knpemidg/solver.py). */
int knp_allreduce_sum(knp_ctx* ctx, double* values, int n);

#ifdef __cplusplus
}
#endif
#endif
