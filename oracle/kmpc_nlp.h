/*
 * oracle/kmpc_nlp.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64) of the nonlinear program that the reference
 * builds in scripts/mpc_utils/MKZMPCPathFollower.jl and hands to Ipopt.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The shipped path (mkz_mpc_path_follower_amd/) never
 * links, imports or calls it.
 *
 * PARITY UNPINNED: the reference has no tests / golden vectors for this path and
 * its Julia + JuMP + Ipopt stack (versions unpinned, README.md:15-24) cannot run
 * in the build container.  This restatement is pinned only by (a) the analytic
 * known-answer cases derivable from the model text, (b) an independent
 * scipy.optimize cross-solve (oracle/make_golden.py) and (c) agreement between
 * two independent solvers in this directory (full-space Ipopt-style IPM in
 * kmpc_ipopt_like.c vs. condensed Newton/IPM in kmpc_condensed.c).
 *
 * Index convention: the Julia model is 1-based (z[1] = initial state,
 * u[1] = first input).  Here k = 0..N for states and k = 0..N-1 for inputs.
 * Inputs are stored interleaved U[2k] = acc_k, U[2k+1] = d_f_k  (accel, steer:
 * the MPC_cmd order, msg/MPC_cmd.msg:2-3).  States X[4k..4k+3] = x, y, psi, v.
 */
#ifndef KMPC_NLP_H
#define KMPC_NLP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kmpc_params {
    int N;              /* horizon                        MKZMPCPathFollower.jl:34 */
    double dt;          /* model discretisation td (s)    :33 */
    double dt_control;  /* control period ts (s)          :28 */
    double L_a, L_b;    /* CoG -> front / rear axle (m)   :31-32 */
    double steer_max, steer_dmax;   /* :41-42 */
    double a_max, a_dmax;           /* :44-45 */
    double v_min, v_max;            /* :47-48 */
    /* cost weights in update_cost() argument order (:158-169):
       C_x, C_y, C_psi, C_v, C_dacc, C_ddf, C_acc, C_df */
    double C[8];
    /* 0 = Cartesian model of MKZMPCPathFollower.jl; 1 = Frenet-frame model of MKZMPCPathFollowerFrenet.jl: states (s, e_y, e_psi, v),
       dynamics :114-123 with the curvature polynomial K(s) (:112), cost :97-103 = the same form with zero references and
       C = (0, C_ey, C_epsi, C_ev, C_dacc, C_ddf, C_acc, C_df) */
    int model;
} kmpc_params;

typedef struct kmpc_problem {
    double z0[4];        /* x0, y0, psi0, v0   update_init_cond :132-138 */
    const double *ref;   /* [(N+1)*3] x_r, y_r, psi_r per stage (stage 0 is a dead input, Q3) */
    double v_target;     /* update_reference 4th arg :146 */
    double u_prev[2];    /* acc_current, d_f_current (update_current_input takes them steer-first, :151) */
    double k_poly[4];    /* Frenet model only: K(s) = k[0] s^3 + k[1] s^2 + k[2] s + k[3]  (highest degree first, Frenet.jl:38-39) */
} kmpc_problem;

/* defaults of MKZMPCPathFollower.jl:28-59 with the given horizon */
void kmpc_params_default(kmpc_params *p, int N);
void kmpc_params_default_frenet(kmpc_params *p, int N);

/* number of decision inputs n = 2N and one-sided inequality rows m = 10N-4 */
int kmpc_n(const kmpc_params *p);
int kmpc_m(const kmpc_params *p);

/* one forward-Euler step of the CoG kinematic bicycle, :115-122 */
void kmpc_step(const kmpc_params *p, const double z[4], const double u[2], double zn[4]);
/* the same for the Frenet model (needs the curvature polynomial) */
void kmpc_step_m(const kmpc_params *p, const double *k_poly, const double z[4], const double u[2], double zn[4]);
void kmpc_stage_jac_m(const kmpc_params *p, const double *k_poly, const double z[4], const double u[2], double A[16], double B[8]);
void kmpc_rollout_m(const kmpc_params *p, const double *k_poly, const double z0[4], const double *U, double *X);
/* stage Jacobians A = df/dz (4x4 row-major), B = df/du (4x2 row-major, columns acc, d_f) */
void kmpc_stage_jac(const kmpc_params *p, const double z[4], const double u[2], double A[16], double B[8]);
/* X[0..3] = z0, X[4(k+1)..] = f(X[4k..], U[2k..]) */
void kmpc_rollout(const kmpc_params *p, const double z0[4], const double *U, double *X);
/* objective :97-103 evaluated on a rollout */
double kmpc_cost(const kmpc_params *p, const kmpc_problem *q, const double *U, const double *X);
/* exact gradient of the state-eliminated objective J(U) by a costate sweep */
void kmpc_grad(const kmpc_params *p, const kmpc_problem *q, const double *U, const double *X, double *g);

/* Inequalities of the state-eliminated problem, all linear in U:  Aineq U <= b.
 * Row families, in this order (m = 10N-4):
 *   [0,      2N)        u_j <= ub_j                      bounds :71-72
 *   [2N,     4N)       -u_j <= -lb_j
 *   [4N,     4N+2(N-1))  rate upper: first step vs u_prev with dt_control (:76,:83),
 *                        then k=1..N-2: u_{k+1}-u_k <= dmax*dt (:77-79,:84-86; Q1: k=0 pair is free)
 *                        rows ordered (k, input) with input 0=acc, 1=d_f; row 0/1 = first-step rows
 *   [.., +2(N-1))        rate lower (negated rows)
 *   [.., +N)             v_k <= v_max, k=1..N      (:67 through v_k = v0 + dt*sum acc)
 *   [.., +N)            -v_k <= -v_min
 * A is dense row-major m x n.  relax>0 widens every bound by relax*max(1,|bound|)
 * (Ipopt's bound_relax_factor, default 1e-8).  */
void kmpc_ineq(const kmpc_params *p, const kmpc_problem *q, double relax, double *A, double *b);
/* max_i (A U - b)_i for the UNRELAXED bounds (positive = violated) */
double kmpc_max_violation(const kmpc_params *p, const kmpc_problem *q, const double *U);

/* Certifier: KKT residuals of the state-eliminated NLP at (U, lam>=0).
 * out[0] = ||grad J + A^T lam||_inf, out[1] = max violation (unrelaxed),
 * out[2] = max_i lam_i * max(0, b_i - (AU)_i)  (complementarity), out[3] = min lam, out[4] = J(U) */
void kmpc_certify(const kmpc_params *p, const kmpc_problem *q, const double *U, const double *lam, double out[5]);

#ifdef __cplusplus
}
#endif
#endif
