// Krylov scalars (device resident, one row of KS_N doubles per system) and work-vector bundle.
#pragma once
#include "knpemi_internal.hpp"

enum { KS_RHO = 0, KS_RHO_OLD, KS_ALPHA, KS_BETA, KS_OMEGA, KS_RES, KS_RES0, KS_BNORM, KS_TOL, KS_RNORM, KS_GM_T2, KS_GM_K, KS_N = 12 };   // KS_RNORM: ||b - A x|| in the cell-volume-weighted norm (PCG)
// PCG reuses the slots only BiCGStab / GMRES write (a system runs one method at a time; every *_INIT op rewrites its slots):
//   KS_CG_XA    ||x||_A^2: x0 . A x0 of the initial guess, raised to the sum of the steps' energies  sum_j alpha_j rho_j
//   KS_CG_EST   estimate of the energy-norm error ||x - x_k||_A of the CURRENT iterate (krylov.hip: OP_CG_BETA)
//   KS_CG_SUM   sum_j alpha_j rho_j = ||x_k - x_0||_A^2 + (cross terms vanish for x0 = 0): the Hestenes-Stiefel identity
//   KS_CG_RN0   the true residual norm of the initial guess in the norm of the residual target (reported as res[0])
enum { KS_CG_XA = KS_RHO_OLD, KS_CG_EST = KS_OMEGA, KS_CG_SUM = KS_GM_T2, KS_CG_RN0 = KS_GM_K };

// restarted GMRES (gmres_solve): per system the Hessenberg matrix (column-major, leading dimension m + 1), the Givens rotations, the
// rotated right-hand side g and the solution y of the small least-squares problem live behind the Krylov scalars in knp_ctx::scal
#define KNP_GM_MAX 30
#define KNP_GM_STRIDE ((KNP_GM_MAX + 1) * KNP_GM_MAX + 3 * KNP_GM_MAX + KNP_GM_MAX + 1 + 5)
#define KNP_GM_OFFSET (KNP_MAX_SYS * KS_N + KNP_MAX_SYS * KNP_MAX_RED)      // doubles in front of the GMRES state in knp_ctx::scal

struct KrylovVecs {
    double *x, *b, *coef;                  // unknown, rhs, operator coefficient (kappa | dnphi)
    bjreal* binv;                          // block-Jacobi inverses (fp32 storage), [nsys][nc][nd*nd]
    const uint16_t* bj_idx = nullptr;      // KNP on structured meshes: per-cell entry of the inverse-block table (instead of binv)
    const bjreal* bj_tab = nullptr;        // [entries][nsys][nd*nd]
    double *r, *z, *p, *w;                 // PCG
    double *rhat, *v, *y;                  // BiCGStab extras (t aliases w)
    double* tmp = nullptr;                 // scratch of the Chebyshev block-Jacobi smoother (BiCGStab), or null
    double bj_lmax = 0.0;                  // > 0: lambda_max(Binv A) estimate -> two-step Chebyshev block-Jacobi
    const float* ivol = nullptr;           // [nc] 1 / cell volume: weights of the residual norms (see krylov.hip: weighted norms)
    bool d8 = false;                       // BiCGStab: stop on the order-8 norms of the residual / load densities (krylov.hip) instead of ||.||_w
    double* gm_V = nullptr;                // GMRES: Krylov basis [gm_m + 1][nsys][nc*nd]
    int gm_m = 0;                          // GMRES: restart length (<= KNP_GM_MAX)
    double r_abs = 0.0;                    // PCG: > 0 -> error-controlled stop on the true residual and the energy-norm error estimate (krylov.hip: cg_converged)
};

// sums over the owned cells of the load measure of the stopping tests, per species (krylov.hip: residual_measure); not all-reduced
int load_measure(knp_ctx* c, const double* b, const float* ivol, bool d8, double* out);
int pcg_solve(knp_ctx* c, KrylovVecs& kv, double rtol, double atol, int maxit, int check_every, int* niter, double* res);
int knp_bj_lambda_max(knp_ctx* c, KrylovVecs& kv, int iters, double* out, bool emi = false);
int bicgstab_solve(knp_ctx* c, KrylovVecs& kv, double rtol, double atol, int maxit, int min_it, int check_every, int* niter,
                   double* res);
// right-preconditioned restarted GMRES(m) with the same preconditioner and the same stopping test on the true residual as bicgstab_solve
// (the reference's KNP solver is PETSc GMRES(30), solver.py:684-701)
int gmres_solve(knp_ctx* c, KrylovVecs& kv, double rtol, double atol, int maxit, int min_it, int check_every, int* niter, double* res);
