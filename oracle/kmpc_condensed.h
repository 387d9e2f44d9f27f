/*
 * oracle/kmpc_condensed.h -- TEST INFRASTRUCTURE (see kmpc_nlp.h).  PARITY UNPINNED.
 *
 * CPU statement (fp64, scalar C) of the algorithm the HIP kernels implement, so
 * the device path can be compared against it iterate-for-iterate, and so
 * bench.py has a "port" CPU baseline to time beside the GPU.  It solves the NLP
 * of MKZMPCPathFollower.jl after eliminating the states by forward simulation
 * (all inequalities become linear in U) with Ipopt's
 * published primal-dual barrier method (Waechter & Biegler, Math. Program. 106,
 * 2006): monotone Fiacco-McCormick mu update, fraction-to-the-boundary rule,
 * Armijo back-tracking on the barrier function (the filter degenerates to this
 * because the iterates stay feasible), kappa_Sigma dual safeguard, gradient-based
 * objective scaling and bound relaxation.  Each iteration re-linearises the
 * bicycle model (Gauss-Newton / exact Lagrangian Hessian of the condensed
 * problem) and takes one Newton step on the condensed KKT system
 * (H + A^T Sigma A) du = -(g + A^T mu/s) by dense Cholesky.
 */
#ifndef KMPC_CONDENSED_H
#define KMPC_CONDENSED_H
#include "kmpc_nlp.h"
#ifdef __cplusplus
extern "C" {
#endif

enum { KMPC_OPTIMAL = 0, KMPC_ITERATION_LIMIT = 1, KMPC_INFEASIBLE = 2, KMPC_NUMERICAL_ERROR = 3 };

typedef struct kmpc_opts {
    int max_iter;        /* outer (linearise + factor) iterations */
    double tol;          /* Ipopt tol on the scaled optimality error (1e-8) */
    int hessian;         /* 0 = Gauss-Newton, 1 = exact (falls back to GN when not PD) */
    double mu_init;      /* cold-start barrier parameter (1.0; Ipopt's default is 0.1) */
    double bound_relax;  /* Ipopt bound_relax_factor = 1e-8 */
    int warm;            /* 1: U on entry is a warm start (blended into the interior) */
    double warm_push;    /* weight of the analytic interior point in the blend (1e-5) */
    double warm_mu;      /* mu_init used with a warm start */
    int max_ls;          /* back-tracking steps */
    int mu_strategy;     /* 0 = Ipopt's monotone Fiacco-McCormick default, 1 = Mehrotra predictor-corrector (adaptive),
                            -1 (default) = 1 */
    int indef_strategy;  /* indefinite exact Hessian: 0 = Gauss-Newton fallback (held 2 iterations), 1 = Ipopt-style delta_w shift,
                            2 = hybrid (0 until the second failure, then 1), -1 (default) = 2 */
    int start;           /* cold-start point: 0 (default) = feed-forward guess inside the bounds; 1 = the reference's start, every input 0
                            (MKZMPCPathFollower.jl:65-72), moved inside the bounds where 0 is not strictly feasible */
} kmpc_opts;

typedef struct kmpc_result {
    int status;
    int iters;           /* linearisations == Cholesky factorisations attempted */
    int n_refactor;      /* extra factorisations (exact->GN fallback, regularisation) */
    int n_ls;            /* total back-tracking trial points */
    int n_solves;        /* triangular solve pairs (1 per iteration monotone; 2+ with the predictor) */
    double cost;         /* unscaled objective at the returned U */
    double viol;         /* max inequality violation vs. the unrelaxed bounds */
    double kkt;          /* final scaled optimality error E_0 */
    double mu;
} kmpc_result;

void kmpc_opts_default(kmpc_opts *o);

/* U [2N] in/out, X [(N+1)*4] out (may be NULL), lam [10N-4] out in kmpc_ineq row order (may be NULL) */
int kmpc_condensed_solve(const kmpc_params *p, const kmpc_problem *q, const kmpc_opts *o,
                         double *U, double *X, double *lam, kmpc_result *res);

/* Batch driver used by tests and bench.py's cpu_baseline leg.  Arrays are the
 * same layouts as the C-ABI in include/kmpc.h.  nthreads<=1 runs serially;
 * otherwise problems are split over pthreads.  Returns 0. */
int kmpc_condensed_solve_batch(const kmpc_params *p, const kmpc_opts *o, int B,
                               const double *z0, const double *ref, const double *v_target,
                               const double *u_prev, double *U /*[B,2N] in/out*/,
                               double *X /*[B,(N+1)*4] or NULL*/, int *status, double *cost,
                               double *viol, int *iters, int nthreads);

/* same, with per-problem work counters (either may be NULL): extra factorisations, line-search trial points */
int kmpc_condensed_solve_batch_stats(const kmpc_params *p, const kmpc_opts *o, int B,
                                     const double *z0, const double *ref, const double *v_target,
                                     const double *u_prev, double *U, double *X, int *status, double *cost,
                                     double *viol, int *iters, int *n_refactor, int *n_ls, int nthreads);

/* condensed Hessian (unscaled) and gradient at U, for unit tests of the device kernels.
 * H is n x n row-major.  hessian: 0 GN, 1 exact. */
void kmpc_condense(const kmpc_params *p, const kmpc_problem *q, const double *U, int hessian,
                   double *H, double *g, double *J);

#ifdef __cplusplus
}
#endif
#endif
