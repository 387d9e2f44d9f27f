/*
 * oracle/kmpc_condensed.c -- TEST INFRASTRUCTURE (see kmpc_condensed.h).  PARITY UNPINNED.
 */
#ifndef KMPC_NOISE_ACCEPT
#define KMPC_NOISE_ACCEPT 100.0 /* same value as csrc/kmpc_common.h */
#endif
#include "kmpc_condensed.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

void kmpc_opts_default(kmpc_opts *o)
{
    o->max_iter = 200;
    o->tol = 1e-8;
    o->hessian = 1;
    o->mu_init = 1.0; /* Ipopt default 0.1; 1.0 measured better at N = 8, 20, 50 (mean -5 %, max 45 -> 31 at N = 20) */
    o->bound_relax = 1e-8;
    o->warm = 0;
    o->warm_push = 1e-5;
    o->warm_mu = 1e-6;
    o->max_ls = 40;
    o->mu_strategy = -1;
    o->indef_strategy = -1;
    o->start = 0;
}

/* ---- "forms": the 5N-2 distinct linear forms a_f^T U behind the 10N-4 one-sided rows.
 * f in [0,n): e_f ; [n, n+R): rate forms (R = 2(N-1)) ; [n+R, n+R+N): speed prefix sums.       */
typedef struct {
    int N, n, R, nf;
    double dt;
} forms_t;

static void forms_apply(const forms_t *F, const double *x, double *y)
{
    const int n = F->n, N = F->N;
    for (int j = 0; j < n; ++j) y[j] = x[j];
    double *yr = y + n;
    yr[0] = x[0];
    yr[1] = x[1];
    for (int k = 1; k <= N - 2; ++k)
        for (int j = 0; j < 2; ++j) yr[2 * k + j] = x[2 * (k + 1) + j] - x[2 * k + j];
    double *ys = y + n + F->R;
    double acc = 0.0;
    for (int k = 1; k <= N; ++k) {
        acc += F->dt * x[2 * (k - 1)];
        ys[k - 1] = acc;
    }
}

static void forms_applyT_add(const forms_t *F, const double *w, double *out)
{
    const int n = F->n, N = F->N;
    for (int j = 0; j < n; ++j) out[j] += w[j];
    const double *wr = w + n;
    out[0] += wr[0];
    out[1] += wr[1];
    for (int k = 1; k <= N - 2; ++k)
        for (int j = 0; j < 2; ++j) {
            out[2 * (k + 1) + j] += wr[2 * k + j];
            out[2 * k + j] -= wr[2 * k + j];
        }
    const double *ws = w + n + F->R;
    double suf = 0.0;
    for (int i = N - 1; i >= 0; --i) { /* acc_i appears in v_k for k > i */
        suf += ws[i];                  /* ws[i] is the form of v_{i+1} */
        out[2 * i] += F->dt * suf;
    }
}

static void forms_gram_add(const forms_t *F, const double *w, double *K)
{
    const int n = F->n, N = F->N;
    for (int j = 0; j < n; ++j) K[j * n + j] += w[j];
    const double *wr = w + n;
    K[0] += wr[0];
    K[n + 1] += wr[1];
    for (int k = 1; k <= N - 2; ++k)
        for (int j = 0; j < 2; ++j) {
            const int a = 2 * (k + 1) + j, b = 2 * k + j;
            const double v = wr[2 * k + j];
            K[a * n + a] += v;
            K[b * n + b] += v;
            K[a * n + b] -= v;
            K[b * n + a] -= v;
        }
    const double *ws = w + n + F->R;
    /* S[i] = sum_{k>i} w(v_k) = sum_{t>=i} ws[t];  K[2i][2i'] += dt^2 S[max(i,i')] */
    double *S = (double *)malloc((size_t)N * sizeof(double));
    double suf = 0.0;
    for (int i = N - 1; i >= 0; --i) {
        suf += ws[i];
        S[i] = suf;
    }
    const double dt2 = F->dt * F->dt;
    for (int i = 0; i < N; ++i)
        for (int i2 = 0; i2 < N; ++i2) K[(2 * i) * n + 2 * i2] += dt2 * S[i > i2 ? i : i2];
    free(S);
}

/* per-form upper/lower right-hand sides matching kmpc_ineq()'s rows */
static void forms_bounds(const kmpc_params *p, const kmpc_problem *q, double relax, const forms_t *F,
                         double *bu, double *bl)
{
    const int n = F->n, N = F->N;
#define RLX(x) (relax * fmax(1.0, fabs(x)))
    for (int j = 0; j < n; ++j) {
        const double ub = (j & 1) ? p->steer_max : p->a_max;
        bu[j] = ub + RLX(ub);
        bl[j] = ub + RLX(ub); /* -u_j <= -lb = ub (symmetric box) */
    }
    for (int j = 0; j < 2; ++j) {
        const double d = (j ? p->steer_dmax : p->a_dmax) * p->dt_control;
        bu[n + j] = d + RLX(d) + q->u_prev[j];
        bl[n + j] = d + RLX(d) - q->u_prev[j];
    }
    for (int k = 1; k <= N - 2; ++k)
        for (int j = 0; j < 2; ++j) {
            const double d = (j ? p->steer_dmax : p->a_dmax) * p->dt;
            bu[n + 2 * k + j] = d + RLX(d);
            bl[n + 2 * k + j] = d + RLX(d);
        }
    for (int k = 0; k < N; ++k) {
        bu[n + F->R + k] = p->v_max + RLX(p->v_max) - q->z0[3];
        bl[n + F->R + k] = -p->v_min + RLX(p->v_min) + q->z0[3];
    }
#undef RLX
}

/* second derivatives of the Euler step wrt (psi, v, d_f), contracted with costate lam[4] -> M[3][3] */
static void stage_hess(const kmpc_params *p, const double z[4], const double u[2], const double lam[4],
                       double M[9])
{
    const double r = p->L_b / (p->L_a + p->L_b), dt = p->dt, v = z[3];
    const double cd = cos(u[1]), sd = sin(u[1]);
    const double D = cd * cd + r * r * sd * sd;
    const double b1 = r / D;
    const double b2 = r * (1.0 - r * r) * (2.0 * sd * cd) / (D * D);
    const double beta = atan(r * tan(u[1]));
    const double c = cos(z[2] + beta), s = sin(z[2] + beta), cb = cos(beta), sb = sin(beta);
    const double lx = lam[0], ly = lam[1], lp = lam[2];
    const double pp = lx * (-dt * v * c) + ly * (-dt * v * s);
    const double pv = lx * (-dt * s) + ly * (dt * c);
    const double pd = lx * (-dt * v * c * b1) + ly * (-dt * v * s * b1);
    const double vd = lx * (-dt * s * b1) + ly * (dt * c * b1) + lp * (dt / p->L_b * cb * b1);
    const double dd = lx * (-dt * v * (c * b1 * b1 + s * b2)) + ly * (dt * v * (-s * b1 * b1 + c * b2)) +
                      lp * (dt * v / p->L_b * (-sb * b1 * b1 + cb * b2));
    M[0] = pp; M[1] = pv; M[2] = pd;
    M[3] = pv; M[4] = 0.; M[5] = vd;
    M[6] = pd; M[7] = vd; M[8] = dd;
}

/* Frenet model: second derivatives of the Euler step (MKZMPCPathFollowerFrenet.jl:112-121) wrt q = (s, e_y, e_psi, v, d_f), contracted
 * with the costate lam[4] -> M[5][5] (symmetric).  With g = ds/dt = v cos(phi) D, phi = e_psi + beta(d_f), D = 1/(1 - e_y K(s)):
 *   s+    = s + dt g,   e_y+ = e_y + dt v sin(phi),   e_psi+ = e_psi + dt (v sin(beta)/L_b - g K),   v+ = v + dt acc
 *   M = dt [ (l0 - l2 K) Hess(g) - l2 (grad g grad K^T + grad K grad g^T) - l2 g K'' e_s e_s^T + l1 Hess(v sin phi) + l2 Hess(v sin(beta)/L_b) ] */
static void stage_hess_frenet(const kmpc_params *p, const double *kp, const double z[4], const double u[2], const double lam[4],
                              double M[25])
{
    const double r = p->L_b / (p->L_a + p->L_b), dt = p->dt;
    const double s = z[0], ey = z[1], ep = z[2], v = z[3];
    const double K = ((kp[0] * s + kp[1]) * s + kp[2]) * s + kp[3], K1 = (3.0 * kp[0] * s + 2.0 * kp[1]) * s + kp[2],
                 K2 = 6.0 * kp[0] * s + 2.0 * kp[1];
    const double cd = cos(u[1]), sd = sin(u[1]);
    const double Dn = cd * cd + r * r * sd * sd;
    const double b1 = r / Dn, b2 = r * (1.0 - r * r) * (2.0 * sd * cd) / (Dn * Dn);
    const double beta = atan(r * tan(u[1])), sb = sin(beta), cb = cos(beta);
    const double C = cos(ep + beta), S = sin(ep + beta);
    const double D = 1.0 / (1.0 - ey * K);
    const double Ds = ey * K1 * D * D, De = K * D * D;
    const double Dss = ey * K2 * D * D + 2.0 * ey * K1 * D * Ds, Dse = K1 * D * D + 2.0 * ey * K1 * D * De, Dee = 2.0 * K * D * De;
    /* gradient and Hessian of g over (s, ey, ep, v, d) */
    const double g = v * C * D;
    const double gg[5] = {v * C * Ds, v * C * De, -v * S * D, C * D, -v * S * b1 * D};
    double G[25] = {0};
    G[0] = v * C * Dss; G[1] = v * C * Dse; G[2] = -v * S * Ds; G[3] = C * Ds; G[4] = -v * S * b1 * Ds;
    G[6] = v * C * Dee; G[7] = -v * S * De; G[8] = C * De; G[9] = -v * S * b1 * De;
    G[12] = -v * C * D; G[13] = -S * D; G[14] = -v * C * b1 * D;
    G[18] = 0.0; G[19] = -S * b1 * D;
    G[24] = v * D * (-C * b1 * b1 - S * b2);
    /* Hessians of h = v sin(phi) and w = v sin(beta)/L_b (upper triangle) */
    double Hh[25] = {0}, Hw[25] = {0};
    Hh[12] = -v * S; Hh[13] = C; Hh[14] = -v * S * b1; Hh[19] = C * b1; Hh[24] = v * (-S * b1 * b1 + C * b2);
    Hw[19] = cb * b1 / p->L_b; Hw[24] = v * (-sb * b1 * b1 + cb * b2) / p->L_b;
    const double l0 = lam[0], l1 = lam[1], l2 = lam[2];
    for (int i = 0; i < 5; ++i)
        for (int j = i; j < 5; ++j) {
            double m = (l0 - l2 * K) * G[5 * i + j] + l1 * Hh[5 * i + j] + l2 * Hw[5 * i + j];
            if (i == 0) m -= l2 * K1 * gg[j];          /* grad K grad g^T (row s)   */
            if (j == 0) m -= l2 * K1 * gg[i];          /* grad g grad K^T (column s) */
            if (i == 0 && j == 0) m -= l2 * g * K2;
            M[5 * i + j] = M[5 * j + i] = dt * m;
        }
}

/* one pass: H = Gauss-Newton part (always), S = second-order part (only if S != NULL) */
static void condense_parts(const kmpc_params *p, const kmpc_problem *q, const double *U,
                           double *H, double *S, double *g, double *J)
{
    const int N = p->N, n = 2 * N;
    const int hessian = S != NULL;
    double *X = (double *)malloc((size_t)(N + 1) * 4 * sizeof(double));
    double *G = (double *)calloc((size_t)4 * n, sizeof(double));
    double *P = (double *)calloc((size_t)(N + 2) * 4, sizeof(double));
    double A[16], B[8];
    kmpc_rollout_m(p, q->k_poly, q->z0, U, X);
    if (J) *J = kmpc_cost(p, q, U, X);
    memset(H, 0, (size_t)n * n * sizeof(double));
    if (S) memset(S, 0, (size_t)n * n * sizeof(double));
    memset(g, 0, (size_t)n * sizeof(double));
    if (hessian == 1) { /* costates P[k] = dJ/dz_k, k = N..1 */
        for (int k = N; k >= 1; --k) {
            double *pk = P + 4 * k;
            pk[0] = 2.0 * p->C[0] * (X[4 * k] - q->ref[3 * k]);
            pk[1] = 2.0 * p->C[1] * (X[4 * k + 1] - q->ref[3 * k + 1]);
            pk[2] = 2.0 * p->C[2] * (X[4 * k + 2] - q->ref[3 * k + 2]);
            pk[3] = (k <= N - 1) ? 2.0 * p->C[3] * (X[4 * k + 3] - q->v_target) : 0.0;
            if (k < N) {
                kmpc_stage_jac_m(p, q->k_poly, X + 4 * k, U + 2 * k, A, B);
                for (int j = 0; j < 4; ++j)
                    for (int i = 0; i < 4; ++i) pk[j] += A[4 * i + j] * P[4 * (k + 1) + i];
            }
        }
    }
    for (int k = 0; k < N; ++k) {
        kmpc_stage_jac_m(p, q->k_poly, X + 4 * k, U + 2 * k, A, B);
        if (hessian == 1) {
            /* M over (z, d_f): 5x5; the Cartesian model only has the (psi, v, d_f) block */
            double M[25] = {0};
            if (p->model == 1) stage_hess_frenet(p, q->k_poly, X + 4 * k, U + 2 * k, P + 4 * (k + 1), M);
            else {
                double M3[9];
                stage_hess(p, X + 4 * k, U + 2 * k, P + 4 * (k + 1), M3);
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) M[5 * (i + 2) + (j + 2)] = M3[3 * i + j];
            }
            /* W rows: G_k[0..3], e_{2k+1} */
            const int nc = 2 * k + 2;
            for (int a = 0; a < nc; ++a) {
                const double wa[5] = {G[a], G[n + a], G[2 * n + a], G[3 * n + a], a == 2 * k + 1 ? 1.0 : 0.0};
                for (int b = 0; b < nc; ++b) {
                    const double wb[5] = {G[b], G[n + b], G[2 * n + b], G[3 * n + b], b == 2 * k + 1 ? 1.0 : 0.0};
                    double s = 0.0;
                    for (int i = 0; i < 5; ++i)
                        for (int j = 0; j < 5; ++j) s += wa[i] * M[5 * i + j] * wb[j];
                    S[a * n + b] += s;
                }
            }
        }
        /* G_{k+1} = [A_k G_k | B_k] */
        for (int c = 0; c < 2 * k; ++c) {
            double col[4], out[4];
            for (int i = 0; i < 4; ++i) col[i] = G[i * n + c];
            for (int i = 0; i < 4; ++i) {
                double s = 0.0;
                for (int j = 0; j < 4; ++j) s += A[4 * i + j] * col[j];
                out[i] = s;
            }
            for (int i = 0; i < 4; ++i) G[i * n + c] = out[i];
        }
        for (int i = 0; i < 4; ++i) {
            G[i * n + 2 * k] = B[2 * i];
            G[i * n + 2 * k + 1] = B[2 * i + 1];
        }
        const int ks = k + 1, nc = 2 * ks;
        const double Q[4] = {p->C[0], p->C[1], p->C[2], ks <= N - 1 ? p->C[3] : 0.0};
        const double e[4] = {X[4 * ks] - q->ref[3 * ks], X[4 * ks + 1] - q->ref[3 * ks + 1],
                             X[4 * ks + 2] - q->ref[3 * ks + 2], X[4 * ks + 3] - q->v_target};
        for (int a = 0; a < nc; ++a) {
            double ga = 0.0;
            for (int i = 0; i < 4; ++i) ga += G[i * n + a] * Q[i] * e[i];
            g[a] += 2.0 * ga;
            for (int b = 0; b < nc; ++b) {
                double s = 0.0;
                for (int i = 0; i < 4; ++i) s += G[i * n + a] * Q[i] * G[i * n + b];
                H[a * n + b] += 2.0 * s;
            }
        }
    }
    /* input terms :99-102 */
    for (int k = 0; k < N; ++k)
        for (int j = 0; j < 2; ++j) {
            const int a = 2 * k + j;
            const double Cu = p->C[6 + j], Cd = p->C[4 + j];
            H[a * n + a] += 2.0 * Cu;
            g[a] += 2.0 * Cu * U[a];
            if (k + 1 < N) {
                const int b = a + 2;
                const double d = U[b] - U[a];
                H[a * n + a] += 2.0 * Cd;
                H[b * n + b] += 2.0 * Cd;
                H[a * n + b] -= 2.0 * Cd;
                H[b * n + a] -= 2.0 * Cd;
                g[a] -= 2.0 * Cd * d;
                g[b] += 2.0 * Cd * d;
            }
        }
    free(X);
    free(G);
    free(P);
}

void kmpc_condense(const kmpc_params *p, const kmpc_problem *q, const double *U, int hessian,
                   double *H, double *g, double *J)
{
    const int n = 2 * p->N;
    if (hessian == 1) {
        double *S = (double *)malloc((size_t)n * n * sizeof(double));
        condense_parts(p, q, U, H, S, g, J);
        for (int i = 0; i < n * n; ++i) H[i] += S[i];
        free(S);
    } else {
        condense_parts(p, q, U, H, NULL, g, J);
    }
}

/* in-place lower Cholesky of row-major n x n; returns 0 ok, 1 if a pivot <= tiny */
static int chol(double *K, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = K[j * n + j];
        for (int k = 0; k < j; ++k) d -= K[j * n + k] * K[j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return 1;
        d = sqrt(d);
        K[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = K[i * n + j];
            for (int k = 0; k < j; ++k) s -= K[i * n + k] * K[j * n + k];
            K[i * n + j] = s / d;
        }
    }
    return 0;
}

static void chol_solve(const double *L, int n, double *x)
{
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

/* Well-centred strictly interior point of the (relaxed) polytope; returns 0 or KMPC_INFEASIBLE.
 * First inputs: the point of the first-step interval closest to 0, kept a quarter of the
 * interval's width away from its ends.  Later accelerations steer v_k away from a speed bound
 * it would otherwise sit on (the pair u_1-u_0 is rate-free, Q1); later steering angles are 0. */
/* Strictly feasible, well-centred starting point (the reference starts every primal at 0, `:65-72`, and lets Ipopt push
 * it inside the bounds; after eliminating the states the start must satisfy the speed rows as well).  It is also a
 * first guess of the solution: accelerations approach the reference speed (mean reference spacing / dt) with time
 * constant 1 s, steering approaches the kinematic feed-forward of the reference's mean curvature; both stay inside
 * 60 % of the box and of the rate limits, the first step inside the middle half of its (box, rate vs. u_prev, speed)
 * interval.  Against the all-zero interior point: mean iterations 9.97 -> 8.35 on the synthetic N = 20 draws. */
static int interior_point(const kmpc_params *p, const kmpc_problem *q, double relax, int start, double *Uf)
{
    const int N = p->N;
    const double frac = 0.6, T = 1.0, r = p->L_b / (p->L_a + p->L_b);
    memset(Uf, 0, (size_t)2 * N * sizeof(double));
    /* Q5: the reference bounds v[1], which is pinned to v0 (:67, :113) -- any v0 outside [v_min, v_max] (relaxed like every bound)
       makes its NLP infeasible, even where a first input could bring v_2 back inside */
    if (!(q->z0[3] >= p->v_min - relax * fmax(1.0, fabs(p->v_min)) && q->z0[3] <= p->v_max + relax * fmax(1.0, fabs(p->v_max))))
        return KMPC_INFEASIBLE;
    double len = 0.0;
    /* reference points 1..N only: point 0 is a dead input of the reference NLP (Q3) and must stay one here */
    if (p->model == 0)
        for (int k = 1; k < N; ++k) len += hypot(q->ref[3 * (k + 1)] - q->ref[3 * k], q->ref[3 * (k + 1) + 1] - q->ref[3 * k + 1]);
    const double s0 = q->z0[0];
    /* Frenet model: the reference speed is v_target itself, the curvature K(s0) of the polynomial */
    const double vref = p->model == 0 ? len / ((N - 1) * p->dt) : q->v_target;
    const double kap = p->model == 0 ? (q->ref[3 * N + 2] - q->ref[3 + 2]) / fmax(len, 1e-6)
                                     : ((q->k_poly[0] * s0 + q->k_poly[1]) * s0 + q->k_poly[2]) * s0 + q->k_poly[3];
    const double sb = fmin(fmax(p->L_b * kap, -0.9), 0.9);
    /* start = 1: the reference's own starting point, every input 0 (MKZMPCPathFollower.jl:65-72) -- no feed-forward; what remains below is
       what keeps the point strictly inside the first-step rate interval (anchored at u_prev), the later rate rows and the speed rows */
    const double ffw = start == 1 ? 0.0 : 1.0;
    const double dff = ffw * fmin(fmax(atan(tan(asin(sb)) / r), -frac * p->steer_max), frac * p->steer_max);
    const double aff = ffw * fmin(fmax((vref - q->z0[3]) / T, -frac * p->a_max), frac * p->a_max);
    for (int j = 0; j < 2; ++j) {
        const double ub = j ? p->steer_max : p->a_max;
        const double d0 = (j ? p->steer_dmax : p->a_dmax) * p->dt_control;
        double lo = fmax(-ub - relax * fmax(1.0, ub), q->u_prev[j] - d0 - relax * fmax(1.0, d0));
        double hi = fmin(ub + relax * fmax(1.0, ub), q->u_prev[j] + d0 + relax * fmax(1.0, d0));
        if (j == 0) { /* v_1 = v0 + dt*acc_0 must lie inside the speed bounds */
            lo = fmax(lo, (p->v_min - relax * fmax(1.0, fabs(p->v_min)) - q->z0[3]) / p->dt);
            hi = fmin(hi, (p->v_max + relax * fmax(1.0, fabs(p->v_max)) - q->z0[3]) / p->dt);
        }
        if (!(lo < hi)) return KMPC_INFEASIBLE;
        const double push = 0.25 * (hi - lo);
        Uf[j] = fmin(fmax(j ? dff : aff, lo + push), hi - push);
    }
    /* Later inputs.  Round 4 (out-of-distribution sweep): until then a speed-margin override inside this loop could jump the
       acceleration by more than the rate rows allow (cars below 1 m/s or above 19 m/s that the feed-forward law decelerates / accelerates: 0.2-0.5 % of
       wide-distribution draws started outside a rate row and ended as Error after one iteration).  Now the start is strictly feasible BY CONSTRUCTION
       whenever the first input's interval above is non-empty -- which is exactly when the reference's NLP is feasible (a_1 is not tied to a_0, Q1, and
       zero later accelerations keep v_1):
         * the acceleration law a_k = clamp(v* - v_k, +-0.6 a_max) (time constant 1 s) towards the reference speed clamped INTO the speed interval's
           margin, v* = clamp(vref, v_min + vm, v_max - vm), from k = 1 on, where the jump from a_0 is free (MKZMPCPathFollower.jl:77-79: the rate
           loop starts at i = 2, Q1);
         * v then approaches v* monotonically (|v* - v| shrinks by dt per step or by dt * a where the law is clipped), so the speed rows hold with
           at least the slack v_1 has, and |a_{k+1} - a_k| <= dt |a_k| <= 0.12 a_max < a_dmax dt: every later rate row holds with slack >= 0.18;
         * start = 1 (the reference's all-zero start): a_k = 0, v stays at v_1. */
    const double vm = fmin(1.0, 0.25 * (p->v_max - p->v_min));
    const double dmax_step = frac * p->steer_dmax * p->dt;
    const double vstar = fmin(fmax(vref, p->v_min + vm), p->v_max - vm);
    double v = q->z0[3] + p->dt * Uf[0], dp = Uf[1];
    for (int k = 1; k < N; ++k) {
        const double a = ffw * fmin(fmax((vstar - v) / T, -frac * p->a_max), frac * p->a_max);
        const double d = fmin(fmax(dff, dp - dmax_step), dp + dmax_step);
        Uf[2 * k] = a; Uf[2 * k + 1] = d;
        v += p->dt * a; dp = d;
    }
    return 0;
}

int kmpc_condensed_solve(const kmpc_params *p, const kmpc_problem *q_in, const kmpc_opts *o,
                         double *U, double *X, double *lam_out, kmpc_result *res)
{
    const int N = p->N, n = 2 * N;
    /* The NLP is invariant under a translation of (x, y): solve it in vehicle-centred coordinates.  Recorded paths live
       hundreds of metres from their origin; positions then carry ~1e-13 m of rounding, i.e. ~1e-12 in the cost -- more
       than the Armijo decrease of the last iterations, which then stall at an error of ~1e-6 (closed-loop test). */
    kmpc_problem q_local = *q_in;
    double *ref_local = (double *)malloc((size_t)(N + 1) * 3 * sizeof(double));
    /* (the Frenet model is not translation-invariant in (s, e_y) -- K depends on s -- and its states are small anyway) */
    const double x_off = p->model == 0 ? q_in->z0[0] : 0.0, y_off = p->model == 0 ? q_in->z0[1] : 0.0;
    for (int k = 0; k <= N; ++k) {
        ref_local[3 * k] = q_in->ref[3 * k] - x_off;
        ref_local[3 * k + 1] = q_in->ref[3 * k + 1] - y_off;
        ref_local[3 * k + 2] = q_in->ref[3 * k + 2];
    }
    q_local.z0[0] = q_in->z0[0] - x_off; q_local.z0[1] = q_in->z0[1] - y_off;
    q_local.ref = ref_local;
    const kmpc_problem *q = &q_local;
    forms_t F = {N, n, 2 * (N - 1), 5 * N - 2, p->dt};
    const int nf = F.nf;
    double *mem = (double *)calloc((size_t)(3 * n * n + 12 * n + 14 * nf + (N + 1) * 8), sizeof(double));
    double *H = mem, *Hgn = H + n * n, *K = Hgn + n * n;
    double *g = K + n * n, *ggn = g + n, *rhs = ggn + n, *du = rhs + n, *Ut = du + n, *Uf = Ut + n,
           *rd = Uf + n, *tmpn = rd + n;
    double *bu = tmpn + 5 * n, *bl = bu + nf, *su = bl + nf, *sl = su + nf, *lu = sl + nf, *ll = lu + nf,
           *au = ll + nf, *dlu = au + nf, *dll = dlu + nf, *w = dll + nf, *aut = w + nf, *w2 = aut + nf;
    double *Xl = w2 + 2 * nf, *Xt = Xl + (N + 1) * 4;
    int status = KMPC_ITERATION_LIMIT, iters = 0, n_refac = 0, n_ls = 0, n_solves = 0;
    /* barrier strategy: -1 = default = Mehrotra (validated on seeded draws at N = 8 ... 56 with the safeguards below) */
    const int mu_strategy = o->mu_strategy >= 0 ? o->mu_strategy : 1;
    double mu = o->warm ? o->warm_mu : o->mu_init, err0 = INFINITY, sc = 1.0, J = 0.0;
    const double kappa_eps = 10.0, kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99, kappa_sigma = 1e10,
                 eta_phi = 1e-8, s_max = 100.0;
    /* unscaled duality-gap bound: the cost is within gap_tol * max(1, |J|) of the optimum whatever the objective scaling was.  (5e-7 would save
       one problem in six its last iteration -- mean 7.90 -> 7.77 at N = 20 -- with cost errors still <= 3.5e-8, but on flat problems the first
       input then moves by up to 5e-5 between two implementations that stop one iteration apart: kept at 1e-7, round 3) */
    const double gap_tol = 1e-7;
    const int max_polish = getenv("KMPC_X_POLISH") ? atoi(getenv("KMPC_X_POLISH")) : 1;
    int gn_hold = 0;
    /* after a tiny fraction-to-the-boundary step the barrier floor applies without its cap at the current mean complementarity: a warm start from a wrong
       point (mu = 1e-7, slacks 1e-5 off the bounds) otherwise crawls to the iteration cap in steps of 1e-6 -- mu_cur can never grow (1 of 32 768 wrong-point
       warm starts at N = 8 and at N = 20; with the rule at most 20 / 34 iterations; warm starts from the own solution and cold starts unchanged; 1e-3 already
       costs the own-solution warm starts at N = 20 half an iteration) */
    const double x_unstick = getenv("KMPC_X_UNSTICK") ? atof(getenv("KMPC_X_UNSTICK")) : 1e-4;
    double alpha_last = 1.0;
    /* degenerate complementarity pairs (slack and multiplier vanish together; structurally the last acceleration input, tied to its neighbour by the
       rate cost only): Newton halves both per iteration (x0.375 with the corrector).  A side seen shrinking that way in an accepted (nearly) full step
       enters K -- and the recovery of its multiplier step -- with theta * lambda/s: the step of a double root (s+ = 0.13 s at theta = 0.6; below 0.45
       the corrected step overshoots the bound).  12 x 4096 seeded problems, N = 20: mean iterations 7.45 -> 7.10, E[worst of 4096] 21.7 -> 19.6;
       N = 8: 6.48 -> 6.13 / 13.5 -> 11.4; N = 12: 8.56 -> 7.76; N = 28: 7.70 -> 7.53; N = 50: 8.97 -> 8.84; same minima (costs to 2e-8).
       KMPC_X_DEGEN = 1 switches it off. */
    const double x_degen = getenv("KMPC_X_DEGEN") ? atof(getenv("KMPC_X_DEGEN")) : 0.6;
    double *thu = (double *)malloc((size_t)2 * nf * sizeof(double)), *thl = thu + nf;
    for (int f = 0; f < 2 * nf; ++f) thu[f] = 1.0;
    int *cand = (int *)calloc((size_t)nf, sizeof(int));
    double dw_last = 0.0, dw_spec = 0.0, hmax_prev = 0.0, reg_final = 0.0;
    /* Tuned on the pooled worst-of-4096 statistics of 48 seeded batches (the kernels carry the same values):
       after a failed first trial (= last/3) the shift grows x3 -- back to the one that worked last iteration -- instead of x8;
       in shifted (non-convex) iterations the barrier floor is rd/100 instead of rd/1000.  Experiment overrides: KMPC_X_GROW, KMPC_X_KRDNC. */
    const double dw_grow = getenv("KMPC_X_GROW") ? atof(getenv("KMPC_X_GROW")) : 3.0;
    /* A trial point is also accepted when the predicted decrease of phi_mu is below the noise of its evaluation (100 eps |phi|: the
       cost is a sum of C e^2 terms whose e = x - x_ref carries eps |x|, |x| >> |e|, so phi is resolved to ~50-70 eps |phi|, not to the
       10 eps of Ipopt's slack): near a low-cost optimum the Armijo test otherwise fails on rounding alone, the search backtracks max_ls
       times and takes a 1e-5 step "by luck" -- ~100 wasted roll-outs per solve on ~0.25 % of the problems.  Experiment override: KMPC_X_NOISE. */
    const double x_decay = getenv("KMPC_X_DECAY") ? atof(getenv("KMPC_X_DECAY")) : 3.0;
    const int x_zero_after = getenv("KMPC_X_ZEROAFTER") ? atoi(getenv("KMPC_X_ZEROAFTER")) : 2;
    int n_first_ok = 0, full_prev = 0;
    const int x_leave = getenv("KMPC_X_LEAVE") ? atoi(getenv("KMPC_X_LEAVE")) : 0;
    const double x_dw0 = getenv("KMPC_X_DW0") ? atof(getenv("KMPC_X_DW0")) : 1e-2;   /* experiment: first shift relative to max |sc H_jj| */
    const double x_sigexp = getenv("KMPC_X_SIGEXP") ? atof(getenv("KMPC_X_SIGEXP")) : 3.0;   /* experiment knobs: Mehrotra exponent, fraction to the boundary */
    const double x_tau = getenv("KMPC_X_TAU") ? atof(getenv("KMPC_X_TAU")) : tau_min;
    const int x_gate = getenv("KMPC_X_GATE") ? atoi(getenv("KMPC_X_GATE")) : 2;
    double err_p1 = INFINITY, err_p2 = INFINITY;
    const double k_noise = getenv("KMPC_X_NOISE") ? atof(getenv("KMPC_X_NOISE")) : KMPC_NOISE_ACCEPT;
    const double kappa_rd_nc = getenv("KMPC_X_KRDNC") ? atof(getenv("KMPC_X_KRDNC")) : 40.0;   /* (1e2 until the first-failure switch; re-measured: 10 ... 300) */
    /* 2 = hybrid: Gauss-Newton fallback until the exact Hessian has failed gn_switch times, delta_w shift from then on */
    const int indef_cfg = o->indef_strategy >= 0 ? o->indef_strategy : 2;
    int indef_strategy = indef_cfg == 2 ? 0 : indef_cfg, n_fail = 0;
    /* Round 3: one failure is enough up to N = 28 -- pooled worst-of-4096 iteration-equivalents 25.75 -> 23.52 at N = 20, 22.9 -> 20.6 at N = 12,
       30.5 -> 29.1 at N = 28 with the mean unchanged (+0.1 ... 0.5 %); at N = 50 the mean would rise 4.6 % for 5 % off the tail, and that
       config is throughput-bound, so the long horizons keep two.  Experiment override: KMPC_X_GNSWITCH. */
    const int gn_switch = getenv("KMPC_X_GNSWITCH") ? atoi(getenv("KMPC_X_GNSWITCH")) : ((N >= 32 || o->warm) ? 2 : 1);   /* (a warm start from a poor point begins at mu = 1e-6: shift mode right away stalls there --
                                                                                                        3 of 32768 wrong-point warm starts hit the iteration cap, mean 10.7 -> 12.4 iterations; tools/warm_probe.py) */
    /* Mehrotra safeguards: the barrier target may not drop below (scaled dual infeasibility)/kappa_rd while that exceeds the
       current complementarity (a Gauss-Newton step does not reduce the dual residual the way an LP/QP step does); and the
       corrected direction is only tried at the full fraction-to-the-boundary step */
    const double kappa_rd = getenv("KMPC_X_KRD") ? atof(getenv("KMPC_X_KRD")) : 1e3;
    /* ... but not while the solve is visibly converging: outside shift mode (two failed exact factorisations, or indef_strategy 1 from the
       start) the floor is dropped whenever the optimality error fell in each of the last two iterations.  There it only slowed the end game
       -- mean iterations 7.94 -> 7.45 (N = 20), 9.83 -> 8.97 (N = 50), 6.93 -> 6.48 (N = 8), worst-of-4096 statistics unchanged (pooled batches).
       Without ANY floor outside shift mode a rare problem cycles (N = 50 bench batch, #1010: mu collapses to 1e-8 at error 5, the next step is
       3 % long, and so on to the iteration cap): its error never falls twice in a row, so it keeps the floor.  KMPC_X_GATE: 0 no gate (no floor
       outside shift mode), 1 floor off after a full primal-dual step, 2 (default) the error rule; KMPC_X_KRD_EASY: floor used when the gate is open. */
    const double kappa_rd_easy = getenv("KMPC_X_KRD_EASY") ? atof(getenv("KMPC_X_KRD_EASY")) : 1e300;
    int have_best = 0;
    double *Ubest = (double *)malloc((size_t)(n + 2 * nf) * sizeof(double));
    int n_polish = 0, n_accept = 0, n_tiny = 0, tiny_stop = 0, n_flat = 0;
    double J_prev = 1e300;

    forms_bounds(p, q, o->bound_relax, &F, bu, bl);
    if (interior_point(p, q, o->bound_relax, o->warm ? 0 : o->start, Uf) != 0) {
        status = KMPC_INFEASIBLE; /* e.g. v0 outside [v_min, v_max] (Q5) */
        /* finite, bound-respecting command for callers that ignore status, as the reference node does */
        for (int k = 0; k < N; ++k) {
            U[2 * k] = fmin(fmax(q->u_prev[0], -p->a_max), p->a_max);
            U[2 * k + 1] = fmin(fmax(q->u_prev[1], -p->steer_max), p->steer_max);
        }
        goto finish;
    }
    if (o->warm) {
        /* U <- Uf + theta (U - Uf) with the largest theta<=1 keeping A U < b, shrunk by (1 - warm_push) */
        for (int j = 0; j < n; ++j) du[j] = U[j] - Uf[j];
        forms_apply(&F, Uf, au);
        forms_apply(&F, du, aut);
        double theta = 1.0;
        for (int f = 0; f < nf; ++f) {
            const double s_u = bu[f] - au[f], s_l = bl[f] + au[f];
            if (aut[f] > 0.0) theta = fmin(theta, s_u / aut[f]);
            if (aut[f] < 0.0) theta = fmin(theta, s_l / -aut[f]);
        }
        theta *= (1.0 - o->warm_push);
        for (int j = 0; j < n; ++j) U[j] = Uf[j] + theta * du[j];
    } else {
        memcpy(U, Uf, (size_t)n * sizeof(double));
    }
    forms_apply(&F, U, au);
    for (int f = 0; f < nf; ++f) {
        su[f] = bu[f] - au[f];
        sl[f] = bl[f] + au[f];
    }

    for (int it = 0; it < o->max_iter; ++it) {
        /* linearise */
        const int exact_h = o->hessian == 1;
        condense_parts(p, q, U, Hgn, exact_h ? H : NULL, g, &J);
        if (exact_h) for (int i = 0; i < n * n; ++i) H[i] += Hgn[i]; /* H = GN + second-order */
        if (it == 0) {
            /* Ipopt gradient-based scaling: nlp_scaling_max_gradient = 100 */
            double gmax = 0.0;
            for (int j = 0; j < n; ++j) gmax = fmax(gmax, fabs(g[j]));
            sc = gmax > 100.0 ? 100.0 / gmax : 1.0;
            for (int f = 0; f < nf; ++f) { lu[f] = mu / su[f]; ll[f] = mu / sl[f]; }
        }
        ++iters;
        /* dual residual r_d = sc*grad J + A^T lam */
        for (int j = 0; j < n; ++j) rd[j] = sc * g[j];
        for (int f = 0; f < nf; ++f) w[f] = lu[f] - ll[f];
        forms_applyT_add(&F, w, rd);
        double rdmax = 0.0, lsum = 0.0, cmax0 = 0.0, gap = 0.0;
        for (int j = 0; j < n; ++j) rdmax = fmax(rdmax, fabs(rd[j]));
        for (int f = 0; f < nf; ++f) {
            lsum += lu[f] + ll[f];
            gap += su[f] * lu[f] + sl[f] * ll[f];
            cmax0 = fmax(cmax0, fmax(su[f] * lu[f], sl[f] * ll[f]));
        }
        const double s_d = fmax(s_max, lsum / (2.0 * nf)) / s_max;
        err_p2 = err_p1; err_p1 = err0;
        err0 = fmax(rdmax / s_d, cmax0 / s_d);
        /* Ipopt's scaled test, plus an UNSCALED duality-gap bound so that the cost is within
           gap_tol*max(1,|J|) of the optimum whatever the objective scaling was */
        const double gap_lim = gap_tol * fmax(1.0, fabs(J));
        /* termination: Ipopt's test (+ the gap bound, pursued for at most max_polish further iterations once
           Ipopt's test has been met: below mu ~ 1e-11 round-off defeats the line search), or Ipopt's
           "acceptable level": error <= acceptable_tol (100*tol) for acceptable_iter (15) iterations in a row */
        if (err0 <= o->tol) { /* last iterate passing Ipopt's test (with its multipliers, for the certifier) */
            memcpy(Ubest, U, (size_t)n * sizeof(double));
            memcpy(Ubest + n, lu, (size_t)nf * sizeof(double));
            memcpy(Ubest + n + nf, ll, (size_t)nf * sizeof(double));
            have_best = 1;
        }
        if (err0 <= o->tol) {
            if (gap / sc <= gap_lim || n_polish >= max_polish) { status = KMPC_OPTIMAL; break; }
            ++n_polish;
        } else if (n_polish > 0 && ++n_polish > max_polish) { status = KMPC_OPTIMAL; break; }
        n_accept = err0 <= 100.0 * o->tol ? n_accept + 1 : 0;
        if (n_accept >= 15) { status = KMPC_OPTIMAL; break; }
        /* rounding floor (the fp32 kernels meet it on large-cost problems: the dual residual is a difference of terms ~1e4 and never
           settles below 100*tol): the objective has not moved by more than 20 eps |J| for 12 iterations in a row -- the arithmetic
           cannot improve the iterate; Optimal if the error is within 1e3 tol (cf. the tiny-step rule below and Ipopt's
           acceptable_obj_change_tol) */
        n_flat = fabs(J - J_prev) <= 20.0 * 2.2e-16 * fmax(1.0, fabs(J)) ? n_flat + 1 : 0;
        J_prev = J;
        if (n_flat >= 12 && err0 <= 1e3 * o->tol) { status = KMPC_OPTIMAL; break; }
        const double mu_min = fmax(o->tol * 1e-2, fmin(o->tol / 10.0, 0.1 * gap_lim * sc / (2.0 * nf)));
        /* monotone barrier update (Ipopt eq. (7)) */
        for (; mu_strategy == 0;) {
            double cmu = 0.0;
            for (int f = 0; f < nf; ++f)
                cmu = fmax(cmu, fmax(fabs(su[f] * lu[f] - mu), fabs(sl[f] * ll[f] - mu)));
            const double errmu = fmax(rdmax / s_d, cmu / s_d);
            if (errmu <= kappa_eps * mu && mu > mu_min) mu = fmax(mu_min, fmin(kappa_mu * mu, pow(mu, theta_mu)));
            else break;
        }
        /* K = sc*H + A^T Sigma A ; rhs = -(sc*g + A^T(mu/s_u - mu/s_l)) */
        /* Indefinite exact Hessian (K not positive definite):
           indef_strategy 0: drop the second-order terms (Gauss-Newton) for this and the next gn_hold_k iterations
                             (a failed factorisation costs as much as a good one; N = 20: 11.1 -> 10.2 factorisations per solve);
           indef_strategy 1: Ipopt's inertia correction -- shift the exact Hessian by delta_w*I, delta_w = 1e-2*max|H_jj| the first
                             time (x10 on failure; Ipopt's 1e-4 / x100 lands 1-2 decades above what is needed and costs the slowest
                             N = 20 problems ~10 % more work), later last/3 (x8 on failure); Gauss-Newton only if the shift exceeds 1e2*max|H_jj|.
           indef_strategy 2: hybrid -- 0 until the exact Hessian has failed gn_switch (2) times, then 1 for the rest of the solve
                             (Gauss-Newton ignores negative curvature and leaves a saddle only slowly: N = 20 worst case 63 -> 35).
           Short horizons do best with 2, long ones (N = 50: <= 28 iterations instead of 100-190) with 1. */
        const int gn_hold_k = 2;
        int use_gn = !exact_h || gn_hold > 0;
        if (gn_hold > 0) --gn_hold;
        double reg = 0.0, hmax = 0.0;
        for (int j = 0; j < n; ++j) hmax = fmax(hmax, fabs(sc * H[j * n + j]));
        /* in shift mode the previous iteration's delta_w / 3 is the FIRST trial (Ipopt retries delta_w = 0 first): in a
           non-convex region the delta_w = 0 attempt fails iteration after iteration -- up to 40 % of the factorisations of the
           slowest problems -- while a decaying shift costs nothing near the solution (dropped below 1e-9 * max|H_jj|) */
        if (!use_gn && indef_strategy == 1 && dw_spec > 0.0) {
            reg = dw_spec / x_decay;
            if (reg < 1e-9 * hmax_prev) reg = 0.0;
            /* after two first-trial successes in a row the unshifted matrix is tried first again (KMPC_X_ZEROAFTER = k, 0 = never): a decaying shift
               slows the end game of the solves that left the non-convex region (error x8 per iteration over the last ten iterations of the slowest
               problem of the bench batch).  12 pooled batches: E[worst of 4096] 22.40 -> 21.71 at N = 20, 19.1 -> 18.2 (N = 12), 28.1 -> 27.3 (N = 28),
               39.3 -> 38.4 (N = 50); means unchanged or slightly lower */
            if (x_zero_after > 0 && n_first_ok >= x_zero_after) reg = 0.0;
        }
        if (!use_gn && indef_strategy == 1) hmax_prev = hmax;
        for (int attempt = 0;; ++attempt) {
            const double *Hs = use_gn ? Hgn : H;
            for (int i = 0; i < n * n; ++i) K[i] = sc * Hs[i];
            for (int f = 0; f < nf; ++f) w[f] = thu[f] * lu[f] / su[f] + thl[f] * ll[f] / sl[f];
            forms_gram_add(&F, w, K);
            for (int j = 0; j < n; ++j) K[j * n + j] += reg;
            if (chol(K, n) == 0) {
                if (!use_gn && reg > 0.0) dw_last = reg;
                if (!use_gn) { dw_spec = reg; n_first_ok = attempt == 0 ? n_first_ok + 1 : 0; }
                /* experiment KMPC_X_LEAVE: an unshifted exact factorisation succeeded in shift mode -> the region is convex again: leave shift mode */
                if (x_leave && !use_gn && indef_cfg == 2 && indef_strategy == 1 && reg == 0.0 && attempt == 0) { indef_strategy = 0; n_fail = 0; }
                reg_final = use_gn ? 0.0 : reg / fmax(hmax, 1e-300);
                break;
            }
            ++n_refac;
            if (!use_gn && indef_strategy == 1) {
                if (reg == 0.0) reg = dw_last > 0.0 ? fmax(1e-10 * hmax, dw_last / 3.0) : x_dw0 * hmax;
                else reg *= (dw_last > 0.0 ? dw_grow : 10.0);
                if (reg > 1e2 * hmax) { use_gn = 1; reg = 0.0; }
            } else if (!use_gn) {
                use_gn = 1; gn_hold = gn_hold_k;
                if (indef_cfg == 2 && ++n_fail >= gn_switch) { indef_strategy = 1; gn_hold = 0; }
            }
            else reg = reg == 0.0 ? 1e-8 : reg * 100.0; /* last resort: shift the Gauss-Newton matrix */
            if (attempt > 40) { status = KMPC_NUMERICAL_ERROR; goto finish; }
        }
        /* Mehrotra predictor-corrector (mu_strategy 1; Ipopt's "adaptive mu" family): the affine-scaling step
           (target mu = 0) on the same factorisation probes how much of the complementarity can be removed;
           sigma = (mu_aff/mu_cur)^3 sets this iteration's barrier target, and the second-order term
           ds_aff*dlam_aff corrects the complementarity linearisation. */
        double *corru = tmpn, *corrl = tmpn + nf;  /* 2 nf <= 5 n */
        for (int f = 0; f < nf; ++f) corru[f] = corrl[f] = 0.0;
        if (mu_strategy == 1) {
            for (int j = 0; j < n; ++j) du[j] = -sc * g[j];
            chol_solve(K, n, du);
            ++n_solves;
            forms_apply(&F, du, aut);
            double apa = 1.0, ada = 1.0, mucur = 0.0, muaff = 0.0;
            for (int f = 0; f < nf; ++f) {
                const double dsu = -aut[f], dsl = aut[f];
                const double dlu_ = -lu[f] - thu[f] * lu[f] / su[f] * dsu, dll_ = -ll[f] - thl[f] * ll[f] / sl[f] * dsl;
                if (dsu < 0.0) apa = fmin(apa, -su[f] / dsu);
                if (dsl < 0.0) apa = fmin(apa, -sl[f] / dsl);
                if (dlu_ < 0.0) ada = fmin(ada, -lu[f] / dlu_);
                if (dll_ < 0.0) ada = fmin(ada, -ll[f] / dll_);
                mucur += su[f] * lu[f] + sl[f] * ll[f];
            }
            for (int f = 0; f < nf; ++f) {
                const double dsu = -aut[f], dsl = aut[f];
                const double dlu_ = -lu[f] - thu[f] * lu[f] / su[f] * dsu, dll_ = -ll[f] - thl[f] * ll[f] / sl[f] * dsl;
                muaff += (su[f] + apa * dsu) * (lu[f] + ada * dlu_) + (sl[f] + apa * dsl) * (ll[f] + ada * dll_);
                corru[f] = dsu * dlu_;
                corrl[f] = dsl * dll_;
            }
            mucur /= 2.0 * nf;
            muaff /= 2.0 * nf;
            const double r3 = muaff / mucur, sigma = fmin(1.0, x_sigexp == 3.0 ? r3 * r3 * r3 : pow(r3, x_sigexp));
            mu = fmax(mu_min, sigma * mucur);
            {
                const int stuck = alpha_last < x_unstick;   /* the last accepted step was a tiny fraction-to-the-boundary step */
                const double kap = (!use_gn && reg > 0.0) ? kappa_rd_nc : ((stuck || indef_strategy == 1 || !(x_gate == 1 ? full_prev : (x_gate == 2 ? (err0 < err_p1 && err_p1 < err_p2) : 1))) ? kappa_rd : kappa_rd_easy);
                mu = fmax(mu, fmin(stuck ? 1e300 : mucur, rdmax / s_d / kap));
            }
        }
        const double tau = fmax(x_tau, 1.0 - mu);
        int accepted = 0;
        double alpha = 0.0, ap = 1.0, ad = 1.0;
        for (int pass = 0; pass < 2 && !accepted; ++pass) {
            if (pass == 1) { /* safeguard: the corrected direction need not be a descent direction of phi_mu -> drop the corrector */
                if (mu_strategy != 1) break;
                for (int f = 0; f < nf; ++f) corru[f] = corrl[f] = 0.0;
            }
            for (int j = 0; j < n; ++j) rhs[j] = -sc * g[j];
            for (int f = 0; f < nf; ++f) w[f] = -((mu - corru[f]) / su[f] - (mu - corrl[f]) / sl[f]);
            forms_applyT_add(&F, w, rhs);
            memcpy(du, rhs, (size_t)n * sizeof(double));
            chol_solve(K, n, du);
            ++n_solves;
            forms_apply(&F, du, aut); /* a_f^T du ; ds_u = -aut, ds_l = +aut */
            ap = 1.0; ad = 1.0;
            double gw = 0.0;
            for (int f = 0; f < nf; ++f) {
                const double dsu = -aut[f], dsl = aut[f];
                dlu[f] = (mu - corru[f] - lu[f] * su[f]) / su[f] - thu[f] * lu[f] / su[f] * dsu;
                dll[f] = (mu - corrl[f] - ll[f] * sl[f]) / sl[f] - thl[f] * ll[f] / sl[f] * dsl;
                gw += (mu / su[f] - mu / sl[f]) * aut[f];
                if (dsu < 0.0) ap = fmin(ap, -tau * su[f] / dsu);
                if (dsl < 0.0) ap = fmin(ap, -tau * sl[f] / dsl);
                if (dlu[f] < 0.0) ad = fmin(ad, -tau * lu[f] / dlu[f]);
                if (dll[f] < 0.0) ad = fmin(ad, -tau * ll[f] / dll[f]);
                {   /* shares of the slack and of its multiplier that the full step takes off; the signature of a degenerate pair: both above 0.3 and within
                       0.2 of each other.  (Also requiring the side to be near its bound or its product well above mu changes nothing or costs iterations:
                       7.18 against 7.10 mean at N = 20) */
                    const double qsu = -dsu / su[f], qsl = -dsl / sl[f], qlu = -dlu[f] / lu[f], qll = -dll[f] / ll[f];
                    cand[f] = ((qsu > 0.3 && qlu > 0.3 && fabs(qsu - qlu) < 0.2) ? 1 : 0) | ((qsl > 0.3 && qll > 0.3 && fabs(qsl - qll) < 0.2) ? 2 : 0);
                }
            }
            /* Armijo on phi_mu(U) = sc*J(U) - mu*sum log s along du; d phi/d alpha = (sc*g + A^T(mu/s_u - mu/s_l))^T du */
            double phi0 = sc * J, dphi = gw;
            for (int f = 0; f < nf; ++f) phi0 -= mu * (log(su[f]) + log(sl[f]));
            for (int j = 0; j < n; ++j) dphi += sc * g[j] * du[j];
            alpha = ap;
            for (int l = 0; l < (pass == 0 && mu_strategy == 1 ? 1 : o->max_ls); ++l, alpha *= 0.5) {
                ++n_ls;
                for (int j = 0; j < n; ++j) Ut[j] = U[j] + alpha * du[j];
                kmpc_rollout_m(p, q->k_poly, q->z0, Ut, Xt);
                double phi = sc * kmpc_cost(p, q, Ut, Xt);
                int ok = 1;
                for (int f = 0; f < nf; ++f) {
                    /* slacks are iterates (as in Ipopt), advanced by s -/+ alpha * a_f^T du: recomputing b - a_f^T U
                       would lose 7 digits to cancellation once an active slack is ~1e-9 */
                    const double a = su[f] - alpha * aut[f], b = sl[f] + alpha * aut[f];
                    if (!(a > 0.0) || !(b > 0.0)) { ok = 0; break; }
                    phi -= mu * (log(a) + log(b));
                }
                /* small slack for round-off as in Ipopt (10 * eps * |phi|) */
                if (getenv("KMPC_TRACE_LS")) fprintf(stderr, "   ls it %d pass %d l %d alpha %.3e ok %d phi-phi0 %.3e  alpha*dphi %.3e  (phi0 %.6e, sc %.3e)\n", it, pass, l, alpha, ok, phi - phi0, alpha * dphi, phi0, sc);
                if (ok && (phi - phi0 - 10.0 * 2.2e-16 * fabs(phi0) <= eta_phi * alpha * dphi || (dphi <= 0.0 && -alpha * dphi <= k_noise * 2.2e-16 * fabs(phi0)))) { accepted = 1; break; }
            }
        }
        if (!accepted) { status = err0 <= 100.0 * o->tol ? KMPC_OPTIMAL : KMPC_NUMERICAL_ERROR; break; }  /* acceptable level reached */
        if (getenv("KMPC_TRACE")) fprintf(stderr, "it %3d J %.10g err0 %.3e mu %.2e ap %.3g ad %.3g alpha %.3g rd %.3e comp %.3e gn %d reg/hmax %.2e\n", it, J, err0, mu, ap, ad, alpha, rdmax, cmax0, use_gn, reg / hmax);
        /* Ipopt's tiny-step rule (tiny_step_tol = 10 eps): two accepted steps in a row below 10 eps relative to the iterate mean the
           arithmetic cannot improve it -- stop; Optimal if the error is within 1e3 tol (the rounding floor of the fp32 kernels'
           dual residual sits there), else a numerical error */
        {
            double stepn = 0.0, umax = 1.0;
            for (int j = 0; j < n; ++j) { stepn = fmax(stepn, fabs(alpha * du[j])); umax = fmax(umax, fabs(U[j])); }
            n_tiny = stepn <= 10.0 * 2.2e-16 * umax ? n_tiny + 1 : 0;
        }
        alpha_last = alpha;
        full_prev = alpha >= 1.0 && ad >= 1.0;   /* the accepted step was a full Newton step in the inputs and in the multipliers */
        memcpy(U, Ut, (size_t)n * sizeof(double));
        if (n_tiny >= 2) { status = err0 <= 1e3 * o->tol ? KMPC_OPTIMAL : KMPC_NUMERICAL_ERROR; tiny_stop = 1; break; }
        /* a candidate (marked where the step was computed, below the fraction-to-the-boundary rule) becomes a degenerate pair when the step was accepted
           (nearly) in full */
        if (x_degen > 0.0 && x_degen < 1.0) for (int f = 0; f < nf; ++f) {
            const int full = alpha >= 0.9 && ad >= 0.9;
            thu[f] = (full && (cand[f] & 1)) ? x_degen : 1.0;
            thl[f] = (full && (cand[f] & 2)) ? x_degen : 1.0;
        }
        for (int f = 0; f < nf; ++f) {
            su[f] -= alpha * aut[f];
            sl[f] += alpha * aut[f];
            lu[f] += ad * dlu[f];
            ll[f] += ad * dll[f];
            lu[f] = fmax(fmin(lu[f], kappa_sigma * mu / su[f]), mu / (kappa_sigma * su[f]));
            ll[f] = fmax(fmin(ll[f], kappa_sigma * mu / sl[f]), mu / (kappa_sigma * sl[f]));
        }
    }

finish:
    /* Run-time guard of the slack iterates (same rule as ipm::solve in csrc/kmpc_ipm.h, round 4): the slacks are iterates and the termination test trusts
       them; an offset between a slack and b -/+ a_f^T U, once there, stays (every update is an increment), so one comparison of the LAST iterate's slacks
       with its freshly evaluated forms covers every iterate of the solve, the saved best one included.  Beyond 1e-9 relative: Error, never Optimal. */
    if (status != KMPC_INFEASIBLE) {
        int drifted = 0;
        forms_apply(&F, U, au);
        for (int f = 0; f < nf; ++f) {
            const double lim = 1e-9 * fmax(1.0, fmax(fmax(fabs(bu[f]), fabs(bl[f])), fabs(au[f])));
            if (!(fabs(su[f] - (bu[f] - au[f])) <= lim && fabs(sl[f] - (bl[f] + au[f])) <= lim)) drifted = 1;
        }
        if (drifted) { status = KMPC_NUMERICAL_ERROR; have_best = 0; }
    }
    /* any later trouble (polishing noise, line-search failure, iteration cap) returns the iterate that passed */
    if (have_best && !(status == KMPC_OPTIMAL && err0 <= o->tol) && !tiny_stop) {
        memcpy(U, Ubest, (size_t)n * sizeof(double));
        memcpy(lu, Ubest + n, (size_t)nf * sizeof(double));
        memcpy(ll, Ubest + n + nf, (size_t)nf * sizeof(double));
        status = KMPC_OPTIMAL;
    }
    free(Ubest);
    free(thu);
    free(cand);
    kmpc_rollout_m(p, q->k_poly, q->z0, U, Xl);
    if (X)
        for (int k = 0; k <= N; ++k) {
            X[4 * k] = Xl[4 * k] + x_off; X[4 * k + 1] = Xl[4 * k + 1] + y_off;
            X[4 * k + 2] = Xl[4 * k + 2]; X[4 * k + 3] = Xl[4 * k + 3];
        }
    if (lam_out) {
        /* kmpc_ineq row order */
        const int R = F.R;
        for (int j = 0; j < n; ++j) { lam_out[j] = lu[j] / sc; lam_out[n + j] = ll[j] / sc; }
        for (int r = 0; r < R; ++r) { lam_out[2 * n + r] = lu[n + r] / sc; lam_out[2 * n + R + r] = ll[n + r] / sc; }
        for (int k = 0; k < N; ++k) {
            lam_out[2 * n + 2 * R + k] = lu[n + R + k] / sc;
            lam_out[2 * n + 2 * R + N + k] = ll[n + R + k] / sc;
        }
    }
    if (res) {
        res->status = status;
        res->iters = iters;
        res->n_refactor = n_refac;
        res->n_ls = n_ls;
        res->n_solves = n_solves;
        res->cost = kmpc_cost(p, q, U, Xl);
        res->viol = kmpc_max_violation(p, q, U);
        res->kkt = err0;
        res->mu = getenv("KMPC_X_REGFINAL") ? reg_final : mu;
    }
    free(mem);
    free(ref_local);
    return status;
}

typedef struct {
    const kmpc_params *p;
    const kmpc_opts *o;
    int b0, b1;
    const double *z0, *ref, *vt, *up;
    double *U, *X;
    int *status;
    double *cost, *viol;
    int *iters, *n_refac, *n_ls;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    const int N = j->p->N;
    double *zeros = (double *)calloc((size_t)(N + 1) * 3, sizeof(double));
    for (int b = j->b0; b < j->b1; ++b) {
        kmpc_problem q;
        memset(&q, 0, sizeof q);
        memcpy(q.z0, j->z0 + 4 * (size_t)b, sizeof q.z0);
        if (j->p->model == 1) { /* Frenet: `ref` carries k_poly [B,4]; the cost references are zero */
            memcpy(q.k_poly, j->ref + 4 * (size_t)b, sizeof q.k_poly);
            q.ref = zeros;
        } else q.ref = j->ref + (size_t)b * (N + 1) * 3;
        q.v_target = j->vt[b];
        q.u_prev[0] = j->up[2 * (size_t)b];
        q.u_prev[1] = j->up[2 * (size_t)b + 1];
        kmpc_result r;
        kmpc_condensed_solve(j->p, &q, j->o, j->U + (size_t)b * 2 * N, j->X ? j->X + (size_t)b * (N + 1) * 4 : NULL,
                             NULL, &r);
        if (j->status) j->status[b] = r.status;
        if (j->cost) j->cost[b] = r.cost;
        if (j->viol) j->viol[b] = r.viol;
        if (j->iters) j->iters[b] = r.iters;
        if (j->n_refac) j->n_refac[b] = r.n_refactor;
        if (j->n_ls) j->n_ls[b] = r.n_ls;
    }
    free(zeros);
    return NULL;
}

int kmpc_condensed_solve_batch_stats(const kmpc_params *p, const kmpc_opts *o, int B, const double *z0,
                                     const double *ref, const double *v_target, const double *u_prev, double *U,
                                     double *X, int *status, double *cost, double *viol, int *iters,
                                     int *n_refactor, int *n_ls, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > B) nthreads = B > 0 ? B : 1;
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    job_t *jobs = (job_t *)malloc((size_t)nthreads * sizeof(job_t));
    for (int t = 0; t < nthreads; ++t) {
        job_t j = {p, o, (int)((long long)B * t / nthreads), (int)((long long)B * (t + 1) / nthreads),
                   z0, ref, v_target, u_prev, U, X, status, cost, viol, iters, n_refactor, n_ls};
        jobs[t] = j;
        if (nthreads == 1) worker(&jobs[t]);
        else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}

int kmpc_condensed_solve_batch(const kmpc_params *p, const kmpc_opts *o, int B, const double *z0,
                               const double *ref, const double *v_target, const double *u_prev, double *U,
                               double *X, int *status, double *cost, double *viol, int *iters, int nthreads)
{
    return kmpc_condensed_solve_batch_stats(p, o, B, z0, ref, v_target, u_prev, U, X, status, cost, viol, iters, NULL, NULL, nthreads);
}
