/*
 * include/kmpc.h -- C ABI of the MI355X batched kinematic-bicycle MPC solver.
 *
 * Drop-in boundary for ONE path of govvijaycal/mkz_mpc_path_follower: the per-step
 * nonlinear MPC solve that scripts/mpc_cmd_pub.jl:115-141 delegates to the Julia module
 * scripts/mpc_utils/MKZMPCPathFollower.jl (JuMP model + Ipopt).  The reference has no FFI
 * for this path (it is an in-process Julia module with global state); the entry points
 * below are what a binding for it would bind -- each cites the module function / global
 * it replaces.  INTEGRATION.md shows the Julia `ccall` and Python `ctypes` stubs.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void*
 *     (NULL = the device's default stream, exactly as in HIP; work is ordered on that stream).
 *   - every function returns 0 on success, <0 on error (kmpc_last_error() has the text).
 *   - the caller owns every buffer; the library keeps no pointer after a call returns
 *     (for the async device entry point: after the stream work completes).
 *   - one handle = one device + one stream; calls on a handle are not re-entrant.
 *   - units as in the reference: x,y [m], psi [rad], v [m/s], acc [m/s^2], d_f [rad, tyre angle].
 *   - input pairs are ordered (acc, d_f) = (MPC_cmd.accel_cmd, MPC_cmd.steer_angle_cmd)
 *     (msg/MPC_cmd.msg:2-3).  NOTE the reference's update_current_input() takes them
 *     steer-first (MKZMPCPathFollower.jl:151); the Python mirror keeps that quirk, the C ABI
 *     does not.
 */
#ifndef KMPC_H
#define KMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMPC_ABI_VERSION 8

/* per-problem status, replaces the Symbol returned by solve_model() (MKZMPCPathFollower.jl:176,182) */
enum {
    KMPC_OPTIMAL = 0,          /* :Optimal                                   */
    KMPC_ITERATION_LIMIT = 1,  /* :UserLimit (Ipopt max_cpu_time cap, :29)   */
    KMPC_INFEASIBLE = 2,       /* :Infeasible (e.g. v0 outside [v_min,v_max], Q5) */
    KMPC_NUMERICAL_ERROR = 3   /* :Error                                     */
};

enum { KMPC_F64 = 0, KMPC_F32 = 1 };

/* error codes */
enum {
    KMPC_OK = 0,
    KMPC_ERR_ARG = -1,      /* bad argument / unsupported configuration */
    KMPC_ERR_HIP = -2,      /* a HIP runtime call failed */
    KMPC_ERR_NODEVICE = -3  /* no gfx950 device / code object not loadable */
};

/* Replaces the module-level constants of MKZMPCPathFollower.jl:28-48 and the Ipopt options of :29. */
typedef struct kmpc_config {
    int32_t N;          /* horizon (:34, default 8); supported 2..56.  Compile-time-horizon kernels: one wave per problem for N = 8, 12,
                           16, 20, 24, 28 (N = 8, the reference's own horizon, also FOUR problems per wave: one per 16-lane row); one
                           four-wave workgroup per problem for N = 32, 36, 40, 44, 48, 50; any other horizon runs the generic kernel
                           (runtime N, matrices in LDS) */
    int32_t dtype;      /* KMPC_F64 / KMPC_F32: arithmetic AND device-buffer element type */
    double dt;          /* :33  0.20 */
    double dt_control;  /* :28  0.10 */
    double L_a, L_b;    /* :31-32 */
    double steer_max, steer_dmax;  /* :41-42 */
    double a_max, a_dmax;          /* :44-45 */
    double v_min, v_max;           /* :47-48 */
    /* solver options (Ipopt defaults where an equivalent exists) */
    int32_t max_iter;    /* cap on linearise+factor iterations (deterministic stand-in for max_cpu_time) */
    int32_t hessian;     /* 0 Gauss-Newton, 1 exact condensed Hessian with GN fallback */
    double tol;          /* Ipopt tol (scaled optimality error), 1e-8 */
    double mu_init;      /* barrier parameter of a cold start (default 1.0; Ipopt's is 0.1) */
    double bound_relax;  /* Ipopt bound_relax_factor, 1e-8 */
    double warm_push;    /* weight of the interior point blended into a warm start (default 1e-5) */
    double warm_mu;      /* mu_init used with a warm start (default 1e-6) */
    int32_t max_ls;      /* back-tracking trial points per iteration */
    int32_t kernel_variant; /* 0 = auto: the compile-time-horizon kernel built for N (see N; at N = 8 batches of 1024 problems or more run the
                               four-problems-per-wave kernel), else the generic one; 1 = always the generic kernel; 2 = as 0 but never the
                               four-per-wave kernel (one wave / one workgroup per problem at every batch size) */
    int32_t mu_strategy;    /* barrier update: 0 = Ipopt's default monotone (Fiacco-McCormick), 1 = Mehrotra predictor-corrector
                               (Ipopt's adaptive family; default) */
    int32_t indef_strategy; /* exact Hessian not positive definite: 0 = Gauss-Newton fallback (held for 2 iterations), 1 = Ipopt-style
                               delta_w shift of the exact Hessian, 2 = hybrid (0 until a failure of the exact factorisation -- the first one on cold starts with N <= 28, the second one on
                               longer horizons and on warm starts --, 1 from then on);
                               2 is the default */
    int32_t schedule;       /* order in which the problems of a batch start on the GPU: 0 = index order, 1 (default) = longest
                               predicted first (key = |v0 - reference speed| + 0.3 * net heading change of the reference), which
                               shortens the tail of a launch whose time is set by its slowest problems.  Results do not depend on it.
                               Calls on one handle must be stream-ordered (the permutation workspace belongs to the handle). */
    int32_t model;          /* 0 (default) = MKZMPCPathFollower.jl, Cartesian states (x, y, psi, v); 1 = MKZMPCPathFollowerFrenet.jl,
                               Frenet-frame states (s, e_y, e_psi, v) with a cubic curvature polynomial (kmpc_solve_batch_frenet;
                               horizons N <= 24, and N = 28 with kernel_variant 0: the compile-time-horizon kernels carry the functor
                               for N = 8, 12, ..., 28).  kmpc_create picks the cost defaults of the chosen module. */
    int32_t start;          /* cold-start point of the inputs (a warm start overrides it): 0 (default) = feed-forward guess inside the bounds
                               (accelerations approach the reference speed, steering the curvature feed-forward); 1 = the reference's own
                               start, every primal 0 (MKZMPCPathFollower.jl:65-72, Q9), moved strictly inside the first-step rate interval
                               around u_prev and the speed rows where 0 is not.  The program is non-convex: the two may end in different
                               local minima on a few problems per thousand (tests/test_certify.py counts them). */
} kmpc_config;

typedef struct kmpc_handle kmpc_handle;

int32_t kmpc_abi_version(void);

/* fills the defaults of MKZMPCPathFollower.jl:28-48 for horizon N and element type dtype */
int32_t kmpc_config_default(kmpc_config *cfg, int32_t N, int32_t dtype);

/* Replaces the module load (`import MKZMPCPathFollower`, mpc_cmd_pub.jl:45-47; model build :29-123).
 * device = HIP device ordinal.  Cost weights start at the module defaults (:51-59). */
int32_t kmpc_create(const kmpc_config *cfg, int32_t device, kmpc_handle **out);
int32_t kmpc_destroy(kmpc_handle *h);

/* Replaces update_cost(cx,cy,cp,cv,cda,cdd,ca,cd) (:158-169), same argument order:
 * C_x, C_y, C_psi, C_v, C_dacc, C_ddf, C_acc, C_df. */
int32_t kmpc_set_cost(kmpc_handle *h, const double w[8]);
int32_t kmpc_get_cost(kmpc_handle *h, double w[8]);

/* The hot path.  Replaces, for B independent problems at once,
 *   update_init_cond (:132-138)     -> z0       [B,4]       x0,y0,psi0,v0
 *   update_reference (:142-147)     -> ref      [B,N+1,3]   x_r,y_r,psi_r per stage (mpc_path layout
 *                                                           xs/ys/psis interleaved; stage 0 unused, Q3)
 *                                      v_target [B]
 *   update_current_input (:151-154) -> u_prev   [B,2]       acc_current, d_f_current
 *   solve_model (:173-183)          -> out_u0   [B,2]       acc_opt[1], d_f_opt[1]
 *                                      out_status [B] int32
 *   get_solver_results (:188-207)   -> out_U    [B,N,2]     acc_opt, d_f_opt       (optional, NULL to skip)
 *                                      out_X    [B,N+1,4]   x,y,psi,v predictions  (optional)
 * plus out_cost [B] (objective :97-103 at the returned inputs), out_viol [B] (max violation of
 * the bounds :65-86, <= bound_relax when feasible), out_iters [B] int32 (optional).
 * warm_U [B,N,2] (optional): if warm != 0 it is read as the starting inputs (JuMP re-solves from the
 * previous primal values, Q9); it is always overwritten with the solution when non-NULL.
 * All pointers are DEVICE pointers of the handle's dtype (status/iters int32); the call is
 * asynchronous on `stream`.  Outputs are always written and always finite and within the input
 * bounds, so a caller that ignores status -- as mpc_cmd_pub.jl:121-132 does -- still gets a command. */
int32_t kmpc_solve_batch(kmpc_handle *h, int32_t B, const void *z0, const void *ref,
                         const void *v_target, const void *u_prev, void *warm_U, int32_t warm,
                         void *out_u0, int32_t *out_status, void *out_cost, void *out_viol,
                         int32_t *out_iters, void *out_U, void *out_X, void *stream);

/* The hot path of the Frenet-frame variant (handle created with cfg.model = 1).  Replaces, for B problems,
 *   update_init_cond(s, ey, epsi, vel)      (MKZMPCPathFollowerFrenet.jl:132-138) -> z0      [B,4]  s0, ey0, epsi0, v0
 *   update_reference(path, k_coeffs, v_des) (:142-147)                            -> k_poly  [B,4]  K(s) coefficients, highest
 *                                                                                    degree first (:38-39); v_target [B]
 *   update_current_input / solve_model / get_solver_results as in kmpc_solve_batch; out_X [B,N+1,4] = s, ey, epsi, v.
 * Cost weights: kmpc_set_cost with (0, C_ey, C_epsi, C_ev, C_dacc, C_ddf, C_acc, C_df) -- update_cost (:158-169) has no x slot. */
int32_t kmpc_solve_batch_frenet(kmpc_handle *h, int32_t B, const void *z0, const void *k_poly, const void *v_target,
                                const void *u_prev, void *warm_U, int32_t warm, void *out_u0, int32_t *out_status,
                                void *out_cost, void *out_viol, int32_t *out_iters, void *out_U, void *out_X, void *stream);

/* ---- packed records (ABI v8) -----------------------------------------------------------------------------------------
 * The same hot path on ONE 64-byte-aligned input record and ONE 64-byte output record per problem instead of four input and five output
 * arrays: with the start-order permutation adjacent workgroups read non-adjacent problems, and 8 ... 32-byte entries of separate arrays are
 * fetched as whole lines (counter traffic 3.3 x the algorithmic bytes at B = 262 144, fp32; 2.1 x with records); a record is a whole number of 64-byte lines.
 *   input record (scalars of the handle's dtype):  [0..3] z0 = x, y, psi, v   (update_init_cond, MKZMPCPathFollower.jl:132-138)
 *                                                  [4]    v_target           (update_reference's v_des, :142-147)
 *                                                  [5..6] u_prev = acc, d_f  (update_current_input, :151-154; accel first)
 *                                                  [7]    unused
 *                                                  [8 + 3 k + (0, 1, 2)] x_ref, y_ref, psi_ref of stage k = 0..N  (update_reference)
 *                                                  zero padding up to kmpc_record_bytes(N, dtype)
 *   output record (64 bytes):  u0 = (acc_opt[1], d_f_opt[1]) (solve_model, :179-182), cost, viol (scalars of the dtype), then status and iters (int32).
 * warm_U / out_U / out_X as in kmpc_solve_batch (optional).  Results are bit-identical to kmpc_solve_batch on the same problems. */
int64_t kmpc_record_bytes(int32_t N, int32_t dtype);   /* stride of the input records in bytes (a multiple of 64), -1 for bad arguments */
/* device kernel: the four arrays of kmpc_solve_batch -> packed input records (for callers that hold the arrays) */
int32_t kmpc_pack_records(kmpc_handle *h, int32_t B, const void *z0, const void *ref, const void *v_target, const void *u_prev,
                          void *records, void *stream);
int32_t kmpc_solve_batch_packed(kmpc_handle *h, int32_t B, const void *records, void *warm_U, int32_t warm, void *out_records,
                                void *out_U, void *out_X, void *stream);

/* Same with HOST pointers: copies in, solves, copies out, synchronises.  This is the form the
 * reference's single-problem API (B = 1) maps onto.  Batches of up to 16 problems run on a pinned, device-mapped host buffer of the
 * handle (the kernel reads its inputs from and writes its outputs to host memory: no copy launches) and the call busy-waits on a completion
 * counter in that buffer (a CPU core spins for the length of the solve, ~50 us; falls back to the stream after 2 ms); larger batches are
 * staged through device memory.  Results are bit-identical either way. */
int32_t kmpc_solve_batch_host(kmpc_handle *h, int32_t B, const void *z0, const void *ref,
                              const void *v_target, const void *u_prev, void *warm_U, int32_t warm,
                              void *out_u0, int32_t *out_status, void *out_cost, void *out_viol,
                              int32_t *out_iters, void *out_U, void *out_X);

/* text of the last error on this handle (or of the last failed kmpc_create when h == NULL) */
const char *kmpc_last_error(kmpc_handle *h);

/* ---- diagnostics used by tests/ (device pointers, handle dtype) ------------------------------ */
/* condensed Hessian H [B,2N,2N] (full symmetric, unscaled), gradient g [B,2N], cost J [B] at U [B,N,2] */
int32_t kmpc_debug_condense(kmpc_handle *h, int32_t B, const void *z0, const void *ref,
                            const void *v_target, const void *U, int32_t hessian, void *H, void *g,
                            void *J, void *stream);
/* The KKT pipeline of the kernel kmpc_solve_batch runs for this handle's horizon (compile-time-horizon kernels only: N = 8, 12, ..., 28
 * one wave per problem; N = 32, 36, ..., 48 and 50 four waves): at inputs U [B,N,2], two-sided form weights w [B,5N-2] (>= 0; order: 2N input boxes,
 * 2(N-1) rate forms, N speed forms), objective scaling sc > 0 and shift reg >= 0 it assembles
 *     K = sc * H(U) + A^T diag(w) A + reg * I      (H: exact condensed Hessian if hessian = 1, Gauss-Newton if 0)
 * on the matrix cores, factors it (blocked Cholesky) and solves once through the block substitutions:
 *     K_out [B,2N,2N] full symmetric, g [B,2N] = grad J(U), x [B,2N] = K^-1 (b - sc*g), ok [B] int32 (0: K not positive definite) */
int32_t kmpc_debug_kkt(kmpc_handle *h, int32_t B, const void *z0, const void *ref, const void *v_target, const void *u_prev,
                       const void *U, const void *w, const void *b, double sc, double reg, int32_t hessian, void *K_out, void *g,
                       void *x, int32_t *ok, void *stream);
/* raw v_mfma_{f64,f32}_16x16x4 probe: a[64], b[64] lane operands -> d[64*4] lane-major results */
int32_t kmpc_debug_mfma_probe(kmpc_handle *h, const void *a, const void *b, void *d, void *stream);

/* ---- batched look-ahead waypoints: the step before the solve ------------------------------------
 * Replaces GPSRefTrajectory of scripts/gps_utils/ref_gps_traj.py: the constructor's path arrays
 * (:87-106; projection and arclength are computed by the host, see ref_traj.py) and
 * get_waypoints(X_init, Y_init, yaw_init[, v_target]) (:131-142, 172-218) for B vehicles at once. */
typedef struct kmpc_path kmpc_path;

/* HOST arrays of length M: time stamps, X, Y (m), psi (rad) and cumulative distance of the recorded path
 * (columns 0, 4, 5, 3, 6 of GPSRefTrajectory.trajectory, :106); they are copied to `device`. */
int32_t kmpc_path_create(int32_t device, int32_t M, const double *t, const double *X, const double *Y,
                         const double *psi, const double *cdist, kmpc_path **out);
int32_t kmpc_path_destroy(kmpc_path *p);

/* DEVICE pointers, fp64: pose [B,3] = (X_init, Y_init, yaw_init); v_target [B] (target-velocity mode,
 * waypoints start ONE step ahead, :175) or NULL (time mode, start at the closest point, :191);
 * ref_out [B,horizon+1,3] = (x_ref, y_ref, psi_ref) per stage -- the layout kmpc_solve_batch takes;
 * stop_out [B] int32 (stop_cmd, :182-184); closest_out [B] int32 or NULL (index of the nearest sample).
 * Asynchronous on `stream`. */
int32_t kmpc_waypoints_batch(kmpc_path *p, int32_t B, int32_t horizon, double traj_dt, const double *pose,
                             const double *v_target, double *ref_out, int32_t *stop_out, int32_t *closest_out,
                             void *stream);
const char *kmpc_path_last_error(kmpc_path *p);

/* ---- closed-loop simulator --------------------------------------------------------
 * Replaces, for B simulated vehicles at once, `n_updates` passes of VehicleSimulator._update_vehicle_model
 * (scripts/vehicle_simulator.py:58-107: dynamic bicycle, linear tyres, 10 Euler sub-steps of 1 ms per pass, heading
 * wrapped to [-pi, pi)) including the actuator lag of _update_low_level_control (:109-113).  n_updates = 10 is one
 * 10 Hz control period of mpc_cmd_pub.jl:87.
 *   state [B,8] fp64 DEVICE, in/out: X, Y, psi, vx, vy, wz, acc, df   (the simulator's attributes, :18-34;
 *                                    state_est publishes x=X, y=Y, psi, v=vx, a=acc, df -- :41-48)
 *   cmd   [B,2] fp64 DEVICE: accel_cmd, steer_angle_cmd of MPC_cmd (:52-55)
 * Asynchronous on `stream` (NULL = the device's default stream).  Errors: negative code, text in kmpc_last_error(NULL). */
int32_t kmpc_sim_advance_batch(int32_t device, int32_t B, void *state, const void *cmd, int32_t n_updates, void *stream);

/* ---- command stage of the node's loop, for B vehicles (scripts/mpc_cmd_pub.jl) --------------------------------------
 * What the loop does between solve_model() and the publish: the waypoint helper's stop flag latches (:100-103); a latched vehicle is
 * commanded accel -1.0 / steer 0.0 (:148-153) and keeps its rate-limit anchor; every other vehicle publishes the solver's first input
 * regardless of the solver status (:120-132) and update_current_input() remembers it for the next solve (:140).
 *   u0 [B,2] fp64 DEVICE in: (accel, steer) from kmpc_solve_batch      stop [B] int32 DEVICE in: kmpc_waypoints_batch's flag
 *   stop_latch [B] uint8 DEVICE in/out                                  u_prev [B,2] fp64 DEVICE in/out (acc, steer)
 *   cmd [B,2] fp64 DEVICE out: accel_cmd, steer_angle_cmd of MPC_cmd
 * Asynchronous on `stream`.  Errors as kmpc_sim_advance_batch. */
int32_t kmpc_command_batch(int32_t device, int32_t B, const void *u0, const int32_t *stop, uint8_t *stop_latch, void *u_prev, void *cmd,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif
