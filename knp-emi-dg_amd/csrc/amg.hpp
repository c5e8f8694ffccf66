// CSR levels of the auxiliary-space AMG hierarchy (device pointers).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include <vector>

struct CsrDev {
    int64_t nrows = 0, ncols = 0, nnz = 0;
    int32_t* rowptr = nullptr;
    int32_t* col = nullptr;
    float* val = nullptr;          // fp32 storage (preconditioner data; symmetric entries round identically), fp64 arithmetic
};

struct AmgLevel {
    int64_t n = 0, ncoarse = 0;
    CsrDev A, P, R;
    double* dinv = nullptr;
    double rho = 1.0, cheb_lower = 0.1;
    int cheb_degree = 3;
    double *x = nullptr, *b = nullptr, *r = nullptr, *d0 = nullptr, *d1 = nullptr;
};

struct AmgHierarchy {
    bool ready = false;
    int ncol = 1;                   // right-hand-side columns carried by one V-cycle (level vectors are [ncol][n])
    int64_t ncg = 0;
    int32_t* dg2cg = nullptr;       // [nc*nd] conforming dof of every DG dof
    int32_t *cg_ptr = nullptr, *cg_idx = nullptr;   // CSR list: conforming dof -> DG dofs (owned cells only)
    // tile-wise restriction (amg.hip: k_restrict_tiles / k_restrict_sum): every tile of consecutive owned cells sums, per conforming
    // dof it touches (a "slot"), the tile's DG values out of LDS; the slots of one conforming dof are then added in fixed order
    int tile_cells = 0;
    int64_t ntiles = 0, nslots = 0;
    int32_t *tile_off = nullptr;    // [ntiles + 1] first slot of every tile
    int32_t *slot_ptr = nullptr;    // [nslots + 1] CSR over slot_idx
    uint16_t* slot_idx = nullptr;   // [nc_owned * nd] tile-local DG dof index, grouped by slot
    int32_t *part_ptr = nullptr, *part_idx = nullptr;   // CSR: conforming dof -> its slots
    double* part = nullptr;         // [ncol][nslots] per-tile partial sums
    int part_cols = 0;
    std::vector<AmgLevel> levels;
    float* pinv = nullptr;          // dense pseudo-inverse of the coarsest level (fp32 storage)
    // Partitioned runs: level 0 holds THIS RANK'S rows only (the conforming dofs of its owned cells + of the membrane facets assigned to it,
    // local numbering), its matrix sub-assembled from those cells / facets: products are per-rank partial sums, made consistent at the
    // shared dofs by interface_accumulate (comm.hip); levels >= 1 stay replicated behind one all-reduce of the level-1 residual
    bool dist0 = false;
    double* t0 = nullptr;           // [ncol][n_0] work vector of the distributed level
    void* graph_exec = nullptr;     // hipGraphExec_t of one V-cycle (fixed kernel sequence on fixed buffers; dist0: levels >= 1)
    bool graph_tried = false;
    // Level 0's zero-guess first Chebyshev update comes with stage 2 of the tile-wise restriction (k_restrict_sum) instead of a launch
    // of its own: one process, tile tables, a smoothed level 0 with a coarser level below.  The V-cycle swaps level buffers while it runs
    // (and a captured graph keeps the pointers of its capture), so the restriction writes where THIS V-cycle will read: entry_* = the
    // level-0 buffers at the start of the captured V-cycle (eager V-cycles start from the current ones)
    bool fuse_first0 = false;
    double *entry_x = nullptr, *entry_r = nullptr, *entry_d0 = nullptr;
};

struct knp_ctx;
int amg_vcycle(knp_ctx* c, AmgHierarchy& H, hipStream_t on_stream = nullptr);
// stage 1 of the tile-wise restriction may be done by a fused kernel of krylov.hip: buffer first, the remaining stages afterwards
int amg_restrict_tiles_prepare(knp_ctx* c, AmgHierarchy& H);
int amg_restrict_finish(knp_ctx* c, AmgHierarchy& H, hipStream_t on_stream = nullptr);
// restricts r_dg, or r_dg - ct * t_dg when t_dg is given
int amg_restrict_from_dg(knp_ctx* c, AmgHierarchy& H, const double* r_dg, hipStream_t on_stream = nullptr, int64_t r_stride = 0,
                         const double* t_dg = nullptr, double ct = 0.0);
void amg_free(AmgHierarchy& H);
