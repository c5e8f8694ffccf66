// Internal declarations of libknpemi_hip.so (gfx950 only).  The public C ABI is
// include/knpemi_hip.h; nothing here is visible to callers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include <map>
#include "amg.hpp"

// block-Jacobi inverses are preconditioner data: stored in fp32 (half the bytes of the largest stream of the fused Krylov
// vector kernels), applied in fp64 arithmetic; the EMI blocks are symmetrised before rounding so that PCG keeps an exactly
// symmetric preconditioner
typedef float bjreal;

#define KNP_MAX_IONS 8          // total species incl. the eliminated one
#define KNP_MAX_SYS 7           // solved species (batched KNP systems)
#define KNP_BLOCK 256
#define KNP_MAX_RED 8           // partial sums per block per system in one reduction pass
#define KNP_ODE_FAIL_SLOT (2 * KNP_MAX_SYS)   // word of knp_ctx::status raised by k_ode_step (read back with the solver status)
#define KNP_PECLET_SLOT (2 * KNP_MAX_SYS + 1)     // bits of a float: max over the owned cells of psi max|z| (max - min nodal phi), set by knp_update_dnphi
#define KNP_STATUS_WORDS (2 * KNP_MAX_SYS + 2)

// facet kinds stored in bits 2..3 of the per-(cell, local facet) flag byte
enum : uint32_t { FK_SIPG = 0u, FK_MEMBRANE = 1u, FK_EXTERIOR = 2u, FK_INACTIVE = 3u };
// bits 0..1: local facet index of this facet in the neighbour cell
// bit 4    : this cell is the `plus` (ECS-like, lower tag) side of the oriented normal n_g

struct MeshDev {
    int dim = 0;
    int64_t nv = 0, nc = 0, nc_owned = 0, nf = 0, nmf = 0;
    int64_t c_begin = 0, c_end = 0;   // cell range of the operator-apply launches: [0, nc_owned), or the interior / boundary part of it
    int64_t n_interior = 0;           // owned cells [0, n_interior) have no ghost neighbour (device order: interior first)
    double* coords = nullptr;      // [nv][4] in 3D (padded), [nv][2] in 2D
    int32_t* cells = nullptr;      // [nc][dim+1]
    double* h = nullptr;           // [nc] cell diameters (UFL CellDiameter: longest edge)
    int32_t* nbr = nullptr;        // [nc][dim+1]
    uint32_t* fflag = nullptr;     // [nc] packed 4 x u8 flag bytes
    int32_t* cfacet = nullptr;     // [nc][dim+1] global facet ids
    int32_t* mf = nullptr;         // [nmf][6]: cell_e, cell_i, lf_e, lf_i, facet, owner flag
    uint16_t* cls = nullptr;       // [nc] geometry class of every cell (structured meshes), or null
    double* cls_table = nullptr;   // [ncls][KNP_CLS_STRIDE]
    double* cls_ext = nullptr;     // [ncls][KNP_CLS_EXT]: per facet 8 derived coefficients (abi.hip: knp_set_geometry_classes), read by the halo-staged KNP apply
    int ncls = 0;
    // per-256-cell-block neighbour tables of the halo-staged applies (3D P1; blocks are aligned at multiples of 256 from cell 0)
    int32_t* hb_src = nullptr;     // [nblk][hb_stride]: 4 * neighbour cell + its local facet, one entry per coupled facet whose neighbour lies
                                   // outside the block; -1 behind the block's last entry
    uint16_t* hb_loc = nullptr;    // [nc_owned][4] LDS entry of the neighbour behind facet i: < 256 in-block cell, else 256 + position in the list
    int hb_stride = 0;             // longest list of the blocks [0, hb_long0), rounded up to 8 (0: no tables)
    int64_t hb_long0 = 0;          // first block whose list is longer than one entry per thread (cut cells of a partition), else the block count
};
#define KNP_CLS_STRIDE 36
#define KNP_CLS_EXT 32
#define KNP_HALO_BLK 256
#define KNP_MAX_MAT 16

struct Params {
    int n_ions = 0;                // total species, last one eliminated
    int n_sys = 0;                 // n_ions - 1
    double C_M = 0, dt = 0, F = 0, R = 0, T = 0, C_phi = 0, psi = 0;
    double tau_emi = 0, tau_knp = 0;
    double z[KNP_MAX_IONS] = {0};
    double f_source[KNP_MAX_IONS] = {0};
    int splitting = 1;
};

// one quadrature rule with the reference basis tabulated at its points (DG-p path, tab_dg.hip); device pointers
struct TabRule {
    int nq = 0;
    const double* w = nullptr;     // [nq], sums to 1
    const double* B = nullptr;     // [nloc][nq][nd]        nloc = 1 (cell rule) or dim+1 (one per local facet)
    const double* dB = nullptr;    // [nloc][nq][nd][dim+1] derivatives with respect to the barycentric coordinates
};
#define KNP_TAB_COUNT 11

struct KernelArgsIons {            // small by-value structs for kernels
    int n;
    double z[KNP_MAX_IONS];
};

struct knp_ctx {
    int device = 0;
    int degree = 1;
    int nd = 0;                    // dofs per cell
    hipStream_t stream = nullptr;
    MeshDev m;
    Params p;
    double* D = nullptr;           // [n_ions][nc]
    // D is piecewise constant by subdomain in every reference configuration (make_global of D_sub, solver.py:88-112): when the cells
    // carry <= KNP_MAX_MAT distinct coefficient tuples the halo-staged KNP apply reads a 1-byte material id per cell and per facet
    // neighbour instead of n_sys doubles (set by knp_set_params; nmat = 0 -> general per-cell D, the kernels read c->D)
    uint8_t* mat = nullptr;        // [nc] material id of the cell
    uint8_t* nmat4 = nullptr;      // [nc][4] material id of the neighbour behind every facet (3D)
    double* dtab = nullptr;        // [n_ions][KNP_MAX_MAT]
    int nmat = 0;
    // host copies of what decides a cell's KNP block-Jacobi block on a structured mesh (abi.hip: build_bj_table): geometry class,
    // material (distinct D tuple, any count up to 65535; empty = too many), flag bytes (facet kinds)
    std::vector<uint16_t> h_cls, h_mat;
    std::vector<uint32_t> h_fflag;
    int* halo_ctr = nullptr;       // [2 operators][2 sets][64 queues] block counters of the persistent halo-staged applies (apply_p1.hip)
    int halo_flip[2] = {0, 0};
    double* rho = nullptr;         // [nc]
    double* fsrc = nullptr;        // [n_sys][nc] DG0 source on ECS cells, or null
    // manufactured-solution mode (splitting == 2): constant coupling coefficients + host-integrated data terms
    double* mms_C = nullptr;       // [n_sys][nc]
    double* extra_emi = nullptr;   // [nc*nd]
    double* extra_knp = nullptr;   // [n_sys][nc*nd]
    // DG-p (p >= 2) path: tabulated rules + assembled cell blocks [nc][dim+2][nd][nd] (tab_dg.hip)
    TabRule tab[KNP_TAB_COUNT];
    double* tab_mem[KNP_TAB_COUNT] = {nullptr};
    double* blk_emi = nullptr;
    double* blk_knp = nullptr;     // [n_sys] x the same
    std::map<int, double*> vecs;   // handle -> device pointer
    std::map<int, int64_t> vlen;
    int next_handle = 1;
    // Krylov workspace
    double* partial = nullptr;     // [grid][KNP_MAX_SYS][KNP_MAX_RED]
    int64_t partial_blocks = 0;
    double* scal = nullptr;        // device scalars
    int* status = nullptr;         // device: per system {converged flag, iterations}, then the ODE failure flag
    void* pinned = nullptr;        // host pinned mirror for status/scalars
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int emi_dg_cheb = -1;          // DG-level Chebyshev step of the EMI preconditioner: -1 default (on for degree 1), 0 / 1 (knp_set_emi_dg_smoother)
    int knp_krylov = 0;            // KNP Krylov method: 0 BiCGStab (default), 1 restarted GMRES (knp_set_knp_krylov)
    int gm_restart = 30;
    double* gm_V = nullptr;        // GMRES basis, allocated at the first GMRES solve
    int gm_alloc = 0;              // basis vectors allocated
    int last_it_emi = 0, last_it_knp = 0;   // iteration counts of the previous solves (chunking of the status polls)
    double knp_early = 0.0;                 // knp_knp_early_stop: a residual this factor under the tolerance ends a BiCGStab solve before min_it
    float last_peclet = -1.0f;              // cell Peclet number of the drift seen by the last status poll (< 0: not read yet)
    // auxiliary-space AMG hierarchies: [0] EMI, [1 + k] KNP species k
    std::vector<AmgHierarchy> amg;
    std::vector<hipStream_t> aux_streams;   // one per extra KNP species: their V-cycles run concurrently
    std::vector<hipEvent_t> aux_events;
    hipEvent_t fork_event = nullptr;
    // distributed
    void* comm = nullptr;          // ncclComm_t: reductions + serial halo exchanges, on the context's stream
    void* comm_halo = nullptr;     // ncclComm_t of the overlapped halo exchanges, on halo_stream
    void* shm = nullptr;           // host-staged shared-memory communicator (comm.hip: knp_comm_init_shm), instead of RCCL
    hipStream_t halo_stream = nullptr;
    hipEvent_t halo_ready = nullptr, halo_done = nullptr;
    int rank = 0, nranks = 1;
    bool dist = false;             // communicator active: halo exchanges + all-reduced reductions
    std::vector<int> halo_peer;
    std::vector<int64_t> halo_send_off, halo_send_cnt, halo_recv_off, halo_recv_cnt;
    int32_t* halo_send_idx = nullptr;   // device: owned cell ids to pack, grouped by peer
    double* halo_sendbuf = nullptr;
    int64_t halo_send_total = 0;
    // row-distributed finest conforming level (amg.hip: dist0; knp_amg_interface): the conforming dofs this rank shares with peers.
    // The list of one peer is the same on both sides (ascending global dof), so message p is sent and received with one layout
    std::vector<int> if_peer;
    std::vector<int64_t> if_off, if_cnt;
    int64_t if_total = 0, if_nuniq = 0;
    int32_t* if_idx = nullptr;      // device [if_total]: local conforming dof of every message position, grouped by peer
    int32_t* if_uvtx = nullptr;     // device [if_nuniq]: the distinct shared dofs
    int32_t* if_aptr = nullptr;     // device [if_nuniq + 1]: CSR over if_asrc
    int32_t* if_asrc = nullptr;     // device: message positions of the dof's other owners in ascending rank order, -1 = this rank's own value
    double *if_send = nullptr, *if_recv = nullptr;   // device [if_total * KNP_MAX_SYS]
    // optional in-solver timing of the operator applies (knp_apply_timing): event pairs recorded around every launch
    bool time_applies = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tev[2];   // [0] EMI, [1] KNP
    size_t tev_used[2] = {0, 0};
    std::string err;
};

#define HIPCHK(ctx, call)                                                        \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);      \
            return -2;                                                           \
        }                                                                        \
    } while (0)

// scalars of the KNP operator, by value into the apply kernels
struct KnpArgs {
    int ns;
    double inv_dt, psi, tau;
    double z[KNP_MAX_SYS];
};
// closed-form P1 facet integrals: mass / triple-product entries of a (D-1)-simplex
template <int D> struct FacetConst;
template <> struct FacetConst<3> { static constexpr double mass = 1.0 / 12.0, trip = 1.0 / 60.0; };
template <> struct FacetConst<2> { static constexpr double mass = 1.0 / 6.0, trip = 1.0 / 24.0; };

// ---- launchers implemented in the .hip files --------------------------------------------
int launch_emi_apply(knp_ctx* c, const double* x, const double* kappa, double* y);
int launch_knp_apply(knp_ctx* c, const double* x, const double* dnphi, double* y);
int launch_emi_blockjacobi(knp_ctx* c, const double* kappa, bjreal* binv);
int launch_knp_blockjacobi(knp_ctx* c, const double* dnphi, bjreal* binv);
int launch_dnphi(knp_ctx* c, const double* phi, double* dnphi);
int launch_neighbour_materials(knp_ctx* c);
int apply_variant(knp_ctx* c, int which);
int launch_kappa(knp_ctx* c, const double* cc, const double* celim, double* kappa);
int launch_emi_rhs(knp_ctx* c, const double* cc, const double* celim, const double* phiM,
                   const double* Ich, double* b);
int launch_knp_rhs(knp_ctx* c, const double* cc, const double* cprev, const double* celim,
                   const double* phi, const double* phiM, const double* Ich, double* b);
int launch_step_updates(knp_ctx* c, const double* cc, double* celim, const double* phi,
                        double* phiM, double* E);
int launch_facet_trace(knp_ctx* c, const double* nodal, int side, double* out);

// DG-p path (tab_dg.hip)
int tab_kappa(knp_ctx* c, const double* cc, const double* celim, double* kappa);
int tab_assemble_knp(knp_ctx* c, const double* phi);
int tab_apply(knp_ctx* c, int which, const double* x, double* y);
int tab_block_inverse(knp_ctx* c, int which, bjreal* binv);
int tab_emi_rhs(knp_ctx* c, const double* cc, const double* celim, const double* phiM, const double* Ich, double* b);
int tab_knp_rhs(knp_ctx* c, const double* cc, const double* cprev, const double* celim, const double* phi, const double* phiM,
                const double* Ich, double* b);
int tab_step_updates(knp_ctx* c, const double* cc, double* celim, const double* phi, double* phiM, double* E, bool do_celim);
int tab_facet_trace(knp_ctx* c, const double* nodal, int side, double* out);
void tab_free(knp_ctx* c);
bool p2_assembled();                 // KNP_P2_ASSEMBLED=1: round-1 path (quadrature-assembled cell blocks) instead of the matrix-free applies

// ring-staged P1 applies on structured 3D meshes (apply_ring.hip): loader wave + LDS-DMA ring + consumer waves
bool ring_usable(const knp_ctx* c, int which);       // which: 0 EMI, 1 KNP
// apply_ring_u.hip: the ring-staged applies for 3D P1 meshes WITHOUT geometry classes (geometry from staged vertex coordinates)
int64_t ring_u_cells(knp_ctx* c, int which);         // leading owned cells the unstructured ring covers (0: not usable); builds its tables once
void ring_u_free(knp_ctx* c);
int ring_u_emi_apply(knp_ctx* c, const MeshDev& m, const double* x, const double* kappa, double* y, int reserve_cus);
int ring_u_knp_apply(knp_ctx* c, const MeshDev& m, const double* x, const double* gphi, double* y, const KnpArgs& ka, int reserve_cus);
int ring_emi_apply(knp_ctx* c, const MeshDev& m, const double* x, const double* kappa, double* y, int reserve_cus);
int ring_knp_apply(knp_ctx* c, const MeshDev& m, const double* x, const double* gphi, double* y, const KnpArgs& ka, int reserve_cus);

// matrix-free DG-P2 applies (apply_p2.hip)
int p2_emi_apply(knp_ctx* c, const double* x, const double* kappa, double* y);
int p2_knp_apply(knp_ctx* c, const double* x, const double* phi, double* y);
int p2_block_inverse(knp_ctx* c, int which, const double* coef, bjreal* binv);

int halo_exchange(knp_ctx* c, double* v, int nfields);
// y = A x on the owned cells with the halo exchange of x folded in (which: 0 = EMI, 1 = KNP): without a communicator a plain
// launch; with one, interior cells are computed while the exchange is in flight on the halo stream / communicator (comm.hip)
int dist_apply(knp_ctx* c, int which, double* x, const double* coef, double* y);

int64_t grid_for(int64_t n);
int launch_nernst_only(knp_ctx* c, const double* cc, const double* celim, double* E);
void comm_destroy(knp_ctx* c);
int allreduce_red(knp_ctx* c, double* red, int count);
int allreduce_max(knp_ctx* c, double* host_value);
int allreduce_max_word(knp_ctx* c, int* dev_word);                 // max of a float-bits status word over the ranks, on the solver's stream
int allreduce_sum_host(knp_ctx* c, double* host_values, int n);    // sum of n <= 56 host values over the ranks (synchronous)
// v [ncol / nil][n][nil] holds per-rank partial sums at the shared conforming dofs: afterwards every owner holds the full sum
// (same bits on every owner: added in rank order).  On the context's stream; 2 messages per peer (comm.hip)
int interface_accumulate(knp_ctx* c, double* v, int64_t n, int ncol);
int max_abs_diff(knp_ctx* c, const double* a, const double* b, int nsys, double* out);
int ode_check_failed(knp_ctx* c);   // ode.hip: reads and clears the ODE failure flag (stream idle); sets c->err
