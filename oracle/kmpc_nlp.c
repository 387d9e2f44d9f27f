/*
 * oracle/kmpc_nlp.c -- TEST INFRASTRUCTURE (see kmpc_nlp.h).  PARITY UNPINNED.
 * Restates the NLP of /root/reference/scripts/mpc_utils/MKZMPCPathFollower.jl.
 */
#include "kmpc_nlp.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* MKZMPCPathFollower.jl:28-59 */
void kmpc_params_default(kmpc_params *p, int N)
{
    p->N = N;
    p->dt = 0.20;
    p->dt_control = 0.10;
    p->L_a = 1.108;
    p->L_b = 1.742;
    p->steer_max = 0.5;
    p->steer_dmax = 0.5;
    p->a_max = 1.0;
    p->a_dmax = 1.5;
    p->v_min = 0.0;
    p->v_max = 20.0;
    p->C[0] = 9.0;    /* C_x    :51 */
    p->C[1] = 9.0;    /* C_y    :52 */
    p->C[2] = 10.0;   /* C_psi  :53 */
    p->C[3] = 0.0;    /* C_v    :54 */
    p->C[4] = 100.0;  /* C_dacc :56 */
    p->C[5] = 1000.0; /* C_ddf  :57 */
    p->C[6] = 0.0;    /* C_acc  :58 */
    p->C[7] = 0.0;    /* C_df   :59 */
    p->model = 0;
}

/* defaults of MKZMPCPathFollowerFrenet.jl:27-59: same constants, weights C_ey 9, C_epsi 10, C_ev 0.5, C_dacc 100, C_ddf 1000 */
void kmpc_params_default_frenet(kmpc_params *p, int N)
{
    kmpc_params_default(p, N);
    p->model = 1;
    p->C[0] = 0.0; p->C[1] = 9.0; p->C[2] = 10.0; p->C[3] = 0.5;
}

int kmpc_n(const kmpc_params *p) { return 2 * p->N; }
int kmpc_m(const kmpc_params *p) { return 10 * p->N - 4; }

/* :115  bta = atan(L_b/(L_a+L_b) * tan(d_f));  :119-122 Euler step */
void kmpc_step(const kmpc_params *p, const double z[4], const double u[2], double zn[4])
{
    const double r = p->L_b / (p->L_a + p->L_b);
    const double beta = atan(r * tan(u[1]));
    zn[0] = z[0] + p->dt * (z[3] * cos(z[2] + beta));
    zn[1] = z[1] + p->dt * (z[3] * sin(z[2] + beta));
    zn[2] = z[2] + p->dt * (z[3] / p->L_b * sin(beta));
    zn[3] = z[3] + p->dt * u[0];
}

void kmpc_stage_jac(const kmpc_params *p, const double z[4], const double u[2], double A[16], double B[8])
{
    const double r = p->L_b / (p->L_a + p->L_b);
    const double td = tan(u[1]);
    const double beta = atan(r * td);
    const double dbeta = r * (1.0 + td * td) / (1.0 + r * r * td * td); /* d beta / d d_f */
    const double c = cos(z[2] + beta), s = sin(z[2] + beta);
    const double dt = p->dt, v = z[3];
    memset(A, 0, 16 * sizeof(double));
    memset(B, 0, 8 * sizeof(double));
    A[0] = 1.0;  A[2] = -dt * v * s;  A[3] = dt * c;
    A[5] = 1.0;  A[6] = dt * v * c;   A[7] = dt * s;
    A[10] = 1.0; A[11] = dt / p->L_b * sin(beta);
    A[15] = 1.0;
    B[1] = -dt * v * s * dbeta;
    B[3] = dt * v * c * dbeta;
    B[5] = dt * v / p->L_b * cos(beta) * dbeta;
    B[6] = dt;
}

void kmpc_rollout(const kmpc_params *p, const double z0[4], const double *U, double *X)
{
    memcpy(X, z0, 4 * sizeof(double)); /* :110-113 */
    for (int k = 0; k < p->N; ++k) kmpc_step(p, X + 4 * k, U + 2 * k, X + 4 * (k + 1));
}

/* ---- Frenet-frame model, MKZMPCPathFollowerFrenet.jl:112-123 (z = s, e_y, e_psi, v) ---------------------------- */
static void frenet_step(const kmpc_params *p, const double *kp, const double z[4], const double u[2], double zn[4])
{
    const double r = p->L_b / (p->L_a + p->L_b);
    const double s = z[0], ey = z[1], ep = z[2], v = z[3];
    const double K = ((kp[0] * s + kp[1]) * s + kp[2]) * s + kp[3];          /* :112 */
    const double beta = atan(r * tan(u[1]));                                  /* :113 */
    const double dsdt = v * cos(ep + beta) / (1.0 - ey * K);                  /* :114 */
    zn[0] = s + p->dt * dsdt;                                                 /* :118 */
    zn[1] = ey + p->dt * (v * sin(ep + beta));                                /* :119 */
    zn[2] = ep + p->dt * (v / p->L_b * sin(beta) - dsdt * K);                 /* :120 */
    zn[3] = v + p->dt * u[0];                                                 /* :121 */
}

static void frenet_stage_jac(const kmpc_params *p, const double *kp, const double z[4], const double u[2], double A[16], double B[8])
{
    const double r = p->L_b / (p->L_a + p->L_b), dt = p->dt;
    const double s = z[0], ey = z[1], ep = z[2], v = z[3];
    const double K = ((kp[0] * s + kp[1]) * s + kp[2]) * s + kp[3], Kp = (3.0 * kp[0] * s + 2.0 * kp[1]) * s + kp[2];
    const double td = tan(u[1]), beta = atan(r * td), db = r * (1.0 + td * td) / (1.0 + r * r * td * td);
    const double c = cos(ep + beta), sn = sin(ep + beta), den = 1.0 - ey * K;
    const double dsdt = v * c / den;
    const double ds_s = v * c * ey * Kp / (den * den), ds_ey = v * c * K / (den * den), ds_ep = -v * sn / den, ds_v = c / den,
                 ds_d = -v * sn / den * db;
    memset(A, 0, 16 * sizeof(double));
    memset(B, 0, 8 * sizeof(double));
    A[0] = 1.0 + dt * ds_s; A[1] = dt * ds_ey; A[2] = dt * ds_ep; A[3] = dt * ds_v;
    A[5] = 1.0; A[6] = dt * v * c; A[7] = dt * sn;
    A[8] = dt * (-ds_s * K - dsdt * Kp); A[9] = -dt * ds_ey * K; A[10] = 1.0 - dt * ds_ep * K; A[11] = dt * (sin(beta) / p->L_b - ds_v * K);
    A[15] = 1.0;
    B[1] = dt * ds_d;
    B[3] = dt * v * c * db;
    B[5] = dt * (v / p->L_b * cos(beta) * db - ds_d * K);
    B[6] = dt;
}

void kmpc_step_m(const kmpc_params *p, const double *kp, const double z[4], const double u[2], double zn[4])
{
    if (p->model == 1) frenet_step(p, kp, z, u, zn); else kmpc_step(p, z, u, zn);
}
void kmpc_stage_jac_m(const kmpc_params *p, const double *kp, const double z[4], const double u[2], double A[16], double B[8])
{
    if (p->model == 1) frenet_stage_jac(p, kp, z, u, A, B); else kmpc_stage_jac(p, z, u, A, B);
}
void kmpc_rollout_m(const kmpc_params *p, const double *kp, const double z0[4], const double *U, double *X)
{
    memcpy(X, z0, 4 * sizeof(double));
    for (int k = 0; k < p->N; ++k) kmpc_step_m(p, kp, X + 4 * k, U + 2 * k, X + 4 * (k + 1));
}

/* :97-103.  Julia i=2..N+1 -> k=1..N (position, heading); i=2..N -> k=1..N-1 (speed, Q3) */
double kmpc_cost(const kmpc_params *p, const kmpc_problem *q, const double *U, const double *X)
{
    const int N = p->N;
    double J = 0.0;
    for (int k = 1; k <= N; ++k) {
        const double ex = X[4 * k] - q->ref[3 * k], ey = X[4 * k + 1] - q->ref[3 * k + 1],
                     ep = X[4 * k + 2] - q->ref[3 * k + 2];
        J += p->C[0] * ex * ex + p->C[1] * ey * ey + p->C[2] * ep * ep;
    }
    for (int k = 1; k <= N - 1; ++k) {
        const double ev = X[4 * k + 3] - q->v_target;
        J += p->C[3] * ev * ev;
    }
    for (int k = 0; k < N; ++k) J += p->C[6] * U[2 * k] * U[2 * k] + p->C[7] * U[2 * k + 1] * U[2 * k + 1];
    for (int k = 0; k < N - 1; ++k) {
        const double da = U[2 * (k + 1)] - U[2 * k], dd = U[2 * (k + 1) + 1] - U[2 * k + 1];
        J += p->C[4] * da * da + p->C[5] * dd * dd;
    }
    return J;
}

void kmpc_grad(const kmpc_params *p, const kmpc_problem *q, const double *U, const double *X, double *g)
{
    const int N = p->N;
    double lam[4], A[16], B[8], t[4];
    /* terminal stage k=N: no speed term (Q3) */
    lam[0] = 2.0 * p->C[0] * (X[4 * N] - q->ref[3 * N]);
    lam[1] = 2.0 * p->C[1] * (X[4 * N + 1] - q->ref[3 * N + 1]);
    lam[2] = 2.0 * p->C[2] * (X[4 * N + 2] - q->ref[3 * N + 2]);
    lam[3] = 0.0;
    for (int k = N - 1; k >= 0; --k) {
        kmpc_stage_jac_m(p, q->k_poly, X + 4 * k, U + 2 * k, A, B);
        for (int j = 0; j < 2; ++j) {
            double s = 0.0;
            for (int i = 0; i < 4; ++i) s += B[2 * i + j] * lam[i];
            g[2 * k + j] = s;
        }
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int i = 0; i < 4; ++i) s += A[4 * i + j] * lam[i];
            t[j] = s;
        }
        if (k >= 1) {
            t[0] += 2.0 * p->C[0] * (X[4 * k] - q->ref[3 * k]);
            t[1] += 2.0 * p->C[1] * (X[4 * k + 1] - q->ref[3 * k + 1]);
            t[2] += 2.0 * p->C[2] * (X[4 * k + 2] - q->ref[3 * k + 2]);
            t[3] += 2.0 * p->C[3] * (X[4 * k + 3] - q->v_target);
        }
        memcpy(lam, t, sizeof t);
    }
    for (int k = 0; k < N; ++k) {
        g[2 * k] += 2.0 * p->C[6] * U[2 * k];
        g[2 * k + 1] += 2.0 * p->C[7] * U[2 * k + 1];
        if (k + 1 < N) {
            g[2 * k] -= 2.0 * p->C[4] * (U[2 * (k + 1)] - U[2 * k]);
            g[2 * k + 1] -= 2.0 * p->C[5] * (U[2 * (k + 1) + 1] - U[2 * k + 1]);
        }
        if (k >= 1) {
            g[2 * k] += 2.0 * p->C[4] * (U[2 * k] - U[2 * (k - 1)]);
            g[2 * k + 1] += 2.0 * p->C[5] * (U[2 * k + 1] - U[2 * (k - 1) + 1]);
        }
    }
}

static double relaxed(double bound, double relax) { return relax * fmax(1.0, fabs(bound)); }

void kmpc_ineq(const kmpc_params *p, const kmpc_problem *q, double relax, double *A, double *b)
{
    const int N = p->N, n = 2 * N, m = 10 * N - 4;
    memset(A, 0, (size_t)m * n * sizeof(double));
    int row = 0;
    /* box upper / lower, :71-72 */
    for (int j = 0; j < n; ++j, ++row) {
        const double ub = (j & 1) ? p->steer_max : p->a_max;
        A[row * n + j] = 1.0;
        b[row] = ub + relaxed(ub, relax);
    }
    for (int j = 0; j < n; ++j, ++row) {
        const double lb = (j & 1) ? -p->steer_max : -p->a_max;
        A[row * n + j] = -1.0;
        b[row] = -lb + relaxed(lb, relax);
    }
    /* rate upper then lower.  first step :76,:83 (dt_control, relative to previous command);
       k=1..N-2 :77-79,:84-86 (dt).  The pair (u_1 - u_0) is not constrained (Q1). */
    for (int sgn = 0; sgn < 2; ++sgn) {
        const double sg = sgn ? -1.0 : 1.0;
        for (int j = 0; j < 2; ++j, ++row) {
            const double dmax = (j ? p->steer_dmax : p->a_dmax) * p->dt_control;
            A[row * n + j] = sg;
            /*  sg*(u_0 - u_prev) <= dmax  */
            b[row] = dmax + relaxed(dmax, relax) + sg * q->u_prev[j];
        }
        for (int k = 1; k <= N - 2; ++k)
            for (int j = 0; j < 2; ++j, ++row) {
                const double dmax = (j ? p->steer_dmax : p->a_dmax) * p->dt;
                A[row * n + 2 * (k + 1) + j] = sg;
                A[row * n + 2 * k + j] = -sg;
                b[row] = dmax + relaxed(dmax, relax);
            }
    }
    /* speed bounds on v_k = v0 + dt*sum_{j<k} acc_j, k=1..N  (:67; v_0 is pinned, Q5) */
    for (int k = 1; k <= N; ++k, ++row) {
        for (int j = 0; j < k; ++j) A[row * n + 2 * j] = p->dt;
        b[row] = p->v_max + relaxed(p->v_max, relax) - q->z0[3];
    }
    for (int k = 1; k <= N; ++k, ++row) {
        for (int j = 0; j < k; ++j) A[row * n + 2 * j] = -p->dt;
        b[row] = -p->v_min + relaxed(p->v_min, relax) + q->z0[3];
    }
}

double kmpc_max_violation(const kmpc_params *p, const kmpc_problem *q, const double *U)
{
    const int n = kmpc_n(p), m = kmpc_m(p);
    double *A = (double *)malloc((size_t)m * n * sizeof(double));
    double *b = (double *)malloc((size_t)m * sizeof(double));
    kmpc_ineq(p, q, 0.0, A, b);
    double worst = -INFINITY;
    for (int i = 0; i < m; ++i) {
        double s = -b[i];
        for (int j = 0; j < n; ++j) s += A[i * n + j] * U[j];
        if (s > worst) worst = s;
    }
    free(A);
    free(b);
    return worst;
}

void kmpc_certify(const kmpc_params *p, const kmpc_problem *q, const double *U, const double *lam, double out[5])
{
    const int N = p->N, n = 2 * N, m = 10 * N - 4;
    double *A = (double *)malloc((size_t)m * n * sizeof(double));
    double *b = (double *)malloc((size_t)m * sizeof(double));
    double *X = (double *)malloc((size_t)(N + 1) * 4 * sizeof(double));
    double *g = (double *)malloc((size_t)n * sizeof(double));
    kmpc_ineq(p, q, 0.0, A, b);
    kmpc_rollout_m(p, q->k_poly, q->z0, U, X);
    kmpc_grad(p, q, U, X, g);
    double viol = -INFINITY, comp = 0.0, lmin = INFINITY;
    for (int i = 0; i < m; ++i) {
        double s = -b[i];
        for (int j = 0; j < n; ++j) s += A[i * n + j] * U[j];
        if (s > viol) viol = s;
        const double c = lam[i] * fmax(0.0, -s);
        if (c > comp) comp = c;
        if (lam[i] < lmin) lmin = lam[i];
        for (int j = 0; j < n; ++j) g[j] += A[i * n + j] * lam[i];
    }
    double stat = 0.0;
    for (int j = 0; j < n; ++j) stat = fmax(stat, fabs(g[j]));
    out[0] = stat;
    out[1] = viol;
    out[2] = comp;
    out[3] = lmin;
    out[4] = kmpc_cost(p, q, U, X);
    free(A);
    free(b);
    free(X);
    free(g);
}
