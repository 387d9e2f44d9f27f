// kmpc_ipm.h -- the interior-point method of the compile-time-horizon solver kernels, written ONCE (gfx950 only).
//
// kmpc_fast.hip (one wave per problem) and kmpc_wide.hip (one four-wave workgroup per problem) differ in how they hold and factor the
// condensed KKT matrix -- tiles of one wave vs. tile rows dealt to four waves, reductions inside a wave vs. across waves -- and in nothing
// else.  Everything else lives here, as function templates over the solver type SV (the "back-end"):
//   * the model of scripts/mpc_utils/MKZMPCPathFollower.jl in state-eliminated form: the 5N-2 linear forms behind its inequality rows
//     (:65-86), roll-out and objective (:97-103, :115-122) as wave scans, costates / gradient / stage records, the terminal
//     sensitivities and the O(N^2) adjoint condensing of the Hessian, the start point;
//   * the solve: Ipopt's primal-dual barrier method specialised to linear inequalities as ONE state machine --
//     FIRST / TRIAL / REFACTOR / RESTEP / FINAL -- with termination rules, barrier update (monotone or Mehrotra), inertia
//     correction and line search.  The CPU checker (test infrastructure, never linked here) states the same algorithm in scalar C.
//
// What a back-end SV supplies (all wave-/workgroup-uniform decisions are taken here on values that are identical in every thread):
//   types / constants  real, N_, n, R, nf, NTH (threads per problem), NF (forms per thread), SROWS, GS (stride of the G_N table), LSTR, MODEL_ID,
//                      MSPLIT (stage at which the adjoint condensing recursion is split, 0 = not split); with MSPLIT > 0 also LOW0 (first thread of
//                      the lower half), GMS (stride of the G_M table) and the LDS buffers gmb [3][GMS], pm(c) [2 MSPLIT] (c = 0..3), hm [2 MSPLIT]
//   ids                lane (stage index), vid (index of the input / form slot this thread holds), refresh_ids()
//   LDS pointers       xb, wb, cb, lin, gnb, gb, Lc, cs, pt, cwt, kc
//   problem data       psi0, v0, vt, rx, ry, rp, up(j); x0, y0, kp0..3 for the Frenet functor
//   exchanges          reduce<NS,NM>(sums, non-negative maxima), reduce_flag<NS,NM>(.., all-true flag), max_any(x), rec_writer(),
//                      sum_stages(x) (sum over the lanes of a stage vector), stage_bcast(x, k) (value of stage lane k)
//   corrector terms    cu(i), cl(i)   (references: LDS in the one-wave kernel, registers in the four-wave kernel)
//   model hooks        eval(U[NV], S), linearize(S, exact, g[NV]) [the gradient also stays with the back-end: grad(i)], drop_second_order(),
//                      interior_point(Uf[NV]), forms_apply(x[NV], y[NF]), forms_applyT(w[NF], o[NV]), stage_form_weights(w[NF]),
//                      form_bounds(f, bu, bl), form_relax(f, upper), dim_N() / dim_n() / dim_nf(), stage_t
//   KKT hooks          kkt_factor(sc, reg, want_hmax) [condense + assemble + factor; max |sc H_jj| -> cs[C_HMAX] when asked],
//                      kkt_affine(x[NV]) = K^-1 (-sc g),  kkt_direction(b[NV], x[NV]) = K^-1 (-sc g + b)
//   best iterate       save_best(U[NV]), load_best(U[NV])
// NV = input slots per thread: 1 in the compile-time-horizon back-ends (their model code below is written for one element per thread),
// up to 2 in the generic kernel (kmpc_kernels.hip, run-time horizon, its own model and KKT code), which instantiates only ipm::solve.
#pragma once
#include "kmpc_math.h"

template <typename T> struct StageV {  // lane k: state k / input k at the evaluated point
    T a, d, v, x, y, psi, c, s, sinb, cosb, b1, b2, ex, ey, ep, ev;
    T K, Kp, iden, dsdt;  // Frenet functor only: curvature, dK/ds, 1 / (1 - e_y K), ds/dt at the stage
};


// Kernel-argument scalars (model constants, limits, tolerances, the doubled weights 2*C_i, products like dt^2 that the host computed
// because there is no scalar fp64 ALU) are copied once into a 32-entry LDS table and read from there: the kernarg segment arrives as
// 16-SGPR tuples which the allocator spills and then reloads WHOLE at every use of any member (~600 v_readlane per iteration before).
enum { PT_DT = 0, PT_DTC, PT_RR, PT_DT2, PT_DTL, PT_LB, PT_TOL, PT_GAP_TOL, PT_TOL_X100, PT_TOL_X1000, PT_TOL_D100, PT_TOL_D10,
       PT_STEER_MAX, PT_A_MAX, PT_STEER_DMAX, PT_A_DMAX, PT_W = 16, PT_V_MIN = 24, PT_V_MAX, PT_RELAX, PT_WARM_PUSH, PT_WARM_MU, PT_MU_INIT,
       PT_INV2NF /* 1 / (2 nf): an fp64 literal in the loop would be hoisted into (and spilled from) a VGPR pair */,
       PT_IKRD_NC /* barrier floor relative to the dual infeasibility in shifted (non-convex) iterations */ };
static_assert(PT_IKRD_NC == 31, "the table has 32 entries");
// wave-uniform scalars of the solve that are read once or twice per iteration: parked in LDS (cs[]), not in VGPRs
enum { C_ERR = 0, C_RDS, C_DWL, C_DWS, C_HMAX, C_MUF, C_PHI0, C_DPHI, C_AD, C_J, C_LGS, C_JP, C_EP1, C_EP2 /* optimality error of the last two iterations */, C_AL /* length of the last accepted step */ };

// sizes every compile-time-horizon solver derives from N (n inputs, R rate forms, nf forms; packed lower triangle of K with the rhs row)
#define KMPC_HORIZON_CONSTANTS(N)                                                                                              \
    static constexpr int n = 2 * (N), R = 2 * ((N) - 1), nf = 5 * (N) - 2;                                                       \
    static constexpr int SROWS = ((N) + 1 + 15) / 16; /* 16-lane rows that carry stage data */                                  \
    static constexpr int LC = n * (n + 3) / 2;        /* packed lower triangle + rhs row, column-major */                        \
    /* start of column j minus j, so that element (row i, col j) lives at offc(j) + i */                                         \
    static constexpr int offc(int j) { return j * (n + 1) - j * (j - 1) / 2 - j; }                                               \
    static DEV int offc_rt(int j) { return j * (n + 1) - ((j * (j - 1)) >> 1) - j; }

// The array-shaped hooks ipm::solve calls, for back-ends with ONE input slot per thread (NV = 1: every compile-time-horizon back-end), on top
// of their scalar model / KKT members eval1, linearize1 (gradient left in gb), kkt_affine1, kkt_direction1, save_best1, load_best1.
#define KMPC_IPM_ONE_SLOT_HOOKS                                                                                        \
    static constexpr int NV = 1;                                                                                        \
    typedef StageV<T> stage_t;                                                                                          \
    static constexpr int dim_N() { return N_; }                                                                         \
    static constexpr int dim_n() { return n; }                                                                          \
    static constexpr int dim_nf() { return nf; }                                                                        \
    DEV void form_bounds(int f, T &bu, T &bl) const { ipm::form_bounds(*this, f, bu, bl); }                             \
    DEV T form_relax(int f, bool upper) const { return ipm::form_relax(*this, f, upper); }                              \
    DEV void forms_apply(const T (&x)[1], T (&y)[NF]) { ipm::forms_apply(*this, x[0], y); }                             \
    DEV void forms_applyT(const T (&w)[NF], T (&o)[1]) { o[0] = ipm::forms_applyT(*this, w); }                          \
    DEV void stage_form_weights(const T (&w)[NF]) { ipm::stage_form_weights(*this, w); }                                \
    DEV bool interior_point(T (&Uf)[1]) { return ipm::interior_point(*this, Uf[0]); }                                   \
    DEV T eval(const T (&U)[1], stage_t &S) { return eval1(U[0], S); }                                                  \
    DEV void linearize(const stage_t &S, bool exact, T (&g)[1]) { g[0] = linearize1(S, exact); }                        \
    DEV T grad(int) const { return gb[vid]; }                                                                           \
    DEV void kkt_affine(T (&x)[1]) { x[0] = kkt_affine1(); }                                                            \
    DEV void kkt_direction(const T (&b)[1], T (&x)[1]) { x[0] = kkt_direction1(b[0]); }                                 \
    DEV void save_best(const T (&U)[1]) { save_best1(U[0]); }                                                           \
    DEV void load_best(T (&U)[1]) { U[0] = load_best1(); }


namespace ipm {

// exchange through LDS among the NTH threads of a problem: inside one wave (NTH <= 64: one wave per problem, or a 16-lane row of a wave in the
// four-per-wave back-end, whose rows branch independently -- a workgroup barrier in divergent code would be undefined) program order is enough
template <int NTH> DEV void xsync() { if constexpr (NTH <= 64) { WFENCE(); } else { __syncthreads(); } }

template <typename T> DEV void fill_param_table(T *q, const KP &p, int nf)
{
    q[PT_DT] = (T)p.dt; q[PT_DTC] = (T)p.dtc; q[PT_RR] = (T)p.r; q[PT_DT2] = (T)p.dt2; q[PT_DTL] = (T)p.dt_over_Lb; q[PT_LB] = (T)p.L_b;
    q[PT_TOL] = (T)p.tol; q[PT_GAP_TOL] = (T)p.gap_tol; q[PT_TOL_X100] = (T)p.tol_x100; q[PT_TOL_X1000] = (T)p.tol_x1000;
    q[PT_TOL_D100] = (T)p.tol_d100; q[PT_TOL_D10] = (T)p.tol_d10;
    q[PT_STEER_MAX] = (T)p.steer_max; q[PT_A_MAX] = (T)p.a_max; q[PT_STEER_DMAX] = (T)p.steer_dmax; q[PT_A_DMAX] = (T)p.a_dmax;
    q[PT_W + 0] = (T)p.C2[0]; q[PT_W + 1] = (T)p.C2[1]; q[PT_W + 2] = (T)p.C2[2]; q[PT_W + 3] = (T)p.C2[3];
    q[PT_W + 4] = (T)p.C2[4]; q[PT_W + 5] = (T)p.C2[5]; q[PT_W + 6] = (T)p.C2[6]; q[PT_W + 7] = (T)p.C2[7];
    q[PT_V_MIN] = (T)p.v_min; q[PT_V_MAX] = (T)p.v_max; q[PT_RELAX] = (T)p.relax; q[PT_WARM_PUSH] = (T)p.warm_push;
    q[PT_WARM_MU] = (T)p.warm_mu; q[PT_MU_INIT] = (T)p.mu_init; q[PT_INV2NF] = (T)(1.0 / (2 * nf));
    q[PT_IKRD_NC] = (T)KMPC_IKRD_NC;
}

// ---- the 5N-2 two-sided linear forms a_f^T U behind the 10N-4 one-sided rows of MKZMPCPathFollower.jl:65-86 ------------------------
// f in [0,n): e_f (input boxes :71-72); [n, n+R): rate forms (:75-86, first step against u_prev with dt_control); [n+R, nf): speed prefix sums (:67)
template <class SV> DEV void form_bounds(const SV &s, int f, typename SV::real &bu, typename SV::real &bl)
{
    typedef typename SV::real T;
    constexpr int n = SV::n, R = SV::R, nf = SV::nf;
    const T relax = s.pt[PT_RELAX];
    if (f < n) {
        const T ub = s.pt[(f & 1) ? PT_STEER_MAX : PT_A_MAX];
        bu = bl = ub + relax * fmax((T)1, ub);
    } else if (f < n + R) {
        const int r = f - n, jj = r & 1, kk = r >> 1;
        const T d = s.pt[jj ? PT_STEER_DMAX : PT_A_DMAX] * s.pt[kk == 0 ? PT_DTC : PT_DT];
        const T u = kk == 0 ? s.up(jj) : (T)0;
        bu = d + relax * fmax((T)1, d) + u; bl = d + relax * fmax((T)1, d) - u;
    } else if (f < nf) {
        const T vmax = s.pt[PT_V_MAX], vmin = s.pt[PT_V_MIN];
        bu = vmax + relax * fmax((T)1, fabs(vmax)) - s.v0;
        bl = -vmin + relax * fmax((T)1, fabs(vmin)) + s.v0;
    } else { bu = bl = (T)1; }
}
template <class SV> DEV typename SV::real form_relax(const SV &s, int f, bool upper)
{
    typedef typename SV::real T;
    constexpr int n = SV::n, R = SV::R;
    const T relax = s.pt[PT_RELAX];
    if (f < n) return relax * fmax((T)1, s.pt[(f & 1) ? PT_STEER_MAX : PT_A_MAX]);
    if (f < n + R) { const int r = f - n; return relax * fmax((T)1, s.pt[(r & 1) ? PT_STEER_DMAX : PT_A_DMAX] * s.pt[(r >> 1) == 0 ? PT_DTC : PT_DT]); }
    return relax * fmax((T)1, fabs(s.pt[upper ? PT_V_MAX : PT_V_MIN]));
}

// y_f = a_f^T x   (thread j < n holds x_j; slot i of a thread is form vid + NTH * i)
template <class SV> DEV void forms_apply(SV &s, typename SV::real x, typename SV::real (&y)[SV::NF])
{
    typedef typename SV::real T;
    constexpr int N = SV::N_, n = SV::n, R = SV::R, nf = SV::nf;
    if (s.vid < n) s.xb[s.vid] = x;
    xsync<SV::NTH>();
    T a = s.lane < N ? s.xb[2 * s.lane] : (T)0;
    a = dpp_scan_prefix<SV::SROWS>(a);
    if (s.lane < N) s.cb[s.lane] = a;   // (every wave keeps its own copy of the stage sums: a wave-level fence orders it)
    WFENCE();
#pragma unroll
    for (int i = 0; i < SV::NF; ++i) {
        const int f = s.vid + SV::NTH * i;
        T v = (T)0;
        if (f < n) v = s.xb[f];
        else if (f < n + R) { const int r = f - n; v = r < 2 ? s.xb[r] : s.xb[r + 2] - s.xb[r]; }
        else if (f < nf) v = s.pt[PT_DT] * s.cb[f - n - R];
        y[i] = v;
    }
    xsync<SV::NTH>();
}
template <class SV> DEV void stage_form_weights(SV &s, const typename SV::real (&w)[SV::NF])
{
    typedef typename SV::real T;
    constexpr int N = SV::N_, n = SV::n, R = SV::R, nf = SV::nf;
#pragma unroll
    for (int i = 0; i < SV::NF; ++i) { const int f = s.vid + SV::NTH * i; if (f < nf) s.wb[f] = w[i]; }
    xsync<SV::NTH>();
    T t = s.lane < N ? s.wb[n + R + s.lane] : (T)0;
    t = dpp_scan_suffix<SV::SROWS>(t, s.lane);
    if (s.lane < N) s.cb[s.lane] = t;
    WFENCE();
}
template <class SV> DEV typename SV::real forms_applyT(SV &s, const typename SV::real (&w)[SV::NF])  // returns (A^T w)_j in thread j
{
    typedef typename SV::real T;
    constexpr int n = SV::n, R = SV::R;
    stage_form_weights(s, w);
    T o = (T)0;
    const int j = s.vid;
    if (j < n) {
        o = s.wb[j];
        if (j < 2) o += s.wb[n + j];
        if (j >= 4) o += s.wb[n + j - 2];
        if (j >= 2 && j < R) o -= s.wb[n + j];
        if (!(j & 1)) o += s.pt[PT_DT] * s.cb[j >> 1];
    }
    xsync<SV::NTH>();
    return o;
}

// ---- Cartesian model (MKZMPCPathFollower.jl) -------------------------------------------------------------------------------------------
// roll-out (:115-122 as prefix scans; vehicle-centred coordinates, x0 = y0 = 0) + objective (:97-103) at U (thread j: U_j).  In the
// four-wave kernel every wave evaluates all stages (lane k = stage k), so the stage data and the cost are in every wave without an exchange
template <class SV> DEV typename SV::real eval_cartesian(SV &s, typename SV::real U, StageV<typename SV::real> &S)
{
    typedef typename SV::real T;
    constexpr int N = SV::N_, n = SV::n, SROWS = SV::SROWS;
    const T dt = s.pt[PT_DT], rr_ = s.pt[PT_RR], dtL = s.pt[PT_DTL];
    if (s.vid < n) s.xb[s.vid] = U;
    xsync<SV::NTH>();
    const int k = s.lane;
    const bool st = k < N;
    const T a = st ? s.xb[2 * k] : (T)0, d = st ? s.xb[2 * k + 1] : (T)0;
    const T an = (k + 1 < N) ? s.xb[2 * k + 2] : a, dn = (k + 1 < N) ? s.xb[2 * k + 3] : d;
    if constexpr (SV::NTH == 64) WSYNC();
    S.a = a; S.d = d;
    const T ia = dpp_scan_prefix<SROWS>(a);
    const T v = s.v0 + dt * (ia - a);
    T sd, cd;
    sincos_small(d, &sd, &cd, s.kc);
    const T Dn = cd * cd + rr_ * rr_ * sd * sd;
    const T rs = rsqrt_(Dn);
    S.sinb = rr_ * sd * rs;
    S.cosb = cd * rs;
    const T iD = rs * rs;  // 1 / Dn
    S.b1 = rr_ * iD;
    S.b2 = rr_ * ((T)1 - rr_ * rr_) * ((T)2 * sd * cd) * (iD * iD);
    const T wp = st ? v * S.sinb : (T)0;
    const T ip = dpp_scan_prefix<SROWS>(wp);
    const T psi = s.psi0 + dtL * (ip - wp);
    T sp, cp;
    sincos_mid(psi, &sp, &cp, s.kc);
    S.c = cp * S.cosb - sp * S.sinb;
    S.s = sp * S.cosb + cp * S.sinb;
    const T wx = st ? v * S.c : (T)0, wy = st ? v * S.s : (T)0;
    const T ix = dpp_scan_prefix<SROWS>(wx), iy = dpp_scan_prefix<SROWS>(wy);
    S.x = dt * (ix - wx);
    S.y = dt * (iy - wy);
    S.v = v; S.psi = psi;
    const bool cs = (k >= 1 && k <= N);
    S.ex = cs ? S.x - s.rx : (T)0;
    S.ey = cs ? S.y - s.ry : (T)0;
    S.ep = cs ? psi - s.rp : (T)0;
    S.ev = (k >= 1 && k <= N - 1) ? v - s.vt : (T)0;
    // (the weights are held doubled -- the form every derivative needs; halving the sum is exact)
    const T Cx2 = s.cwt[0], Cy2 = s.cwt[1], Cp2 = s.cwt[2], Cv2 = s.cwt[3], Cda2 = s.cwt[4], Cdd2 = s.cwt[5], Ca2 = s.cwt[6], Cd2 = s.cwt[7];
    T Jl = Cx2 * S.ex * S.ex + Cy2 * S.ey * S.ey + Cp2 * S.ep * S.ep + Cv2 * S.ev * S.ev;
    if (st) Jl += Ca2 * a * a + Cd2 * d * d;
    if (k < N - 1) Jl += Cda2 * (an - a) * (an - a) + Cdd2 * (dn - d) * (dn - d);
    Jl *= (T)0.5;
    const T J = s.sum_stages(Jl);
    if constexpr (SV::NTH > 64) __syncthreads();  // xb is free again
    return J;
}

// costates by suffix scans -> gradient (left in gb, returned: thread j holds g_j); the 13 linearisation scalars of every stage go to the LDS
// records (A02 A03 A12 A13 A23 Bdx Bdy Bdp mpp mpv mpd mvd mdd), the terminal sensitivities G_N [3][GS] to gnb
template <class SV> DEV typename SV::real linearize_cartesian(SV &s, const StageV<typename SV::real> &S, bool exact)
{
    typedef typename SV::real T;
    constexpr int N = SV::N_, n = SV::n, SROWS = SV::SROWS, LSTR = SV::LSTR, GS = SV::GS;
    const T dt = s.pt[PT_DT], dtL = s.pt[PT_DTL];
    const int k = s.lane, lane = s.lane;
    const bool st = k < N;
    const T Cx2 = s.cwt[0], Cy2 = s.cwt[1], Cp2 = s.cwt[2], Cv2 = s.cwt[3], Cda2 = s.cwt[4], Cdd2 = s.cwt[5], Ca2 = s.cwt[6], Cd2 = s.cwt[7];
    const T lx = Cx2 * S.ex, ly = Cy2 * S.ey, lp = Cp2 * S.ep, lv = Cv2 * S.ev;
    const T px = dpp_scan_suffix<SROWS>(lx, lane), py = dpp_scan_suffix<SROWS>(ly, lane);
    const T px1 = dpp_mov0<0x130, 0xf>(px), py1 = dpp_mov0<0x130, 0xf>(py);  // wave_shl:1 -> value of lane+1
    const T A02 = st ? -dt * S.v * S.s : (T)0, A12 = st ? dt * S.v * S.c : (T)0;
    const T A03 = st ? dt * S.c : (T)0, A13 = st ? dt * S.s : (T)0, A23 = st ? dtL * S.sinb : (T)0;
    const T tp = lp + (st ? A02 * px1 + A12 * py1 : (T)0);
    const T pp = dpp_scan_suffix<SROWS>(tp, lane);
    const T pp1 = dpp_mov0<0x130, 0xf>(pp);
    const T tv = lv + (st ? A03 * px1 + A13 * py1 + A23 * pp1 : (T)0);
    const T pv = dpp_scan_suffix<SROWS>(tv, lane);
    const T pv1 = dpp_mov0<0x130, 0xf>(pv);
    const T Bdx = st ? -dt * S.v * S.s * S.b1 : (T)0, Bdy = st ? dt * S.v * S.c * S.b1 : (T)0;
    const T Bdp = st ? dtL * S.v * S.cosb * S.b1 : (T)0;
    const T aprev = dpp_mov0<0x138, 0xf>(S.a), dprev = dpp_mov0<0x138, 0xf>(S.d);  // wave_shr:1 -> lane-1
    const T anext = dpp_mov0<0x130, 0xf>(S.a), dnext = dpp_mov0<0x130, 0xf>(S.d);
    T ga = dt * pv1 + Ca2 * S.a, gd = Bdx * px1 + Bdy * py1 + Bdp * pp1 + Cd2 * S.d;
    if (k >= 1) { ga += Cda2 * (S.a - aprev); gd += Cdd2 * (S.d - dprev); }
    if (k < N - 1) { ga -= Cda2 * (anext - S.a); gd -= Cdd2 * (dnext - S.d); }
    // Terminal sensitivities for condense_adjoint: column j of G_N = Phi(N, k+1) B_k e_j, k = j / 2.  The stage Jacobians are unit
    // upper triangular (x, y <- psi, v; psi <- v), so the transition matrix is made of suffix sums over the later stages:
    //   d psi_N / d v = P3(k) = sum_{s>k} A23_s,   d x_N / d psi = X2(k) = sum_{s>k} A02_s,
    //   d x_N / d v = sum_{s>k} [A03_s + A02_s (P3(k) - R_s)],  R_s = sum_{t>=s} A23_t          (y alike)
    const T Rs = dpp_scan_suffix<SROWS>(A23, lane);
    const T ux = A03 - A02 * Rs, uy = A13 - A12 * Rs;
    const T X2 = dpp_scan_suffix<SROWS>(A02, lane) - A02, Y2 = dpp_scan_suffix<SROWS>(A12, lane) - A12;
    const T zx = dpp_scan_suffix<SROWS>(ux, lane) - ux, zy = dpp_scan_suffix<SROWS>(uy, lane) - uy;
    const T P3 = Rs - A23;
    T mpp = 0, mpv = 0, mpd = 0, mvd = 0, mdd = 0;
    if (exact && st) {
        const T v = S.v, c = S.c, sn = S.s, b1 = S.b1, b2 = S.b2;
        mpp = px1 * (-dt * v * c) + py1 * (-dt * v * sn);
        mpv = px1 * (-dt * sn) + py1 * (dt * c);
        mpd = px1 * (-dt * v * c * b1) + py1 * (-dt * v * sn * b1);
        mvd = px1 * (-dt * sn * b1) + py1 * (dt * c * b1) + pp1 * (dtL * S.cosb * b1);
        mdd = px1 * (-dt * v * (c * b1 * b1 + sn * b2)) + py1 * (dt * v * (-sn * b1 * b1 + c * b2)) +
              pp1 * (dtL * v * (-S.sinb * b1 * b1 + S.cosb * b2));
    }
    // Mid-horizon sensitivities for the split recursion of condense_adjoint: column j of G_M = Phi(M, k+1) B_k e_j, k < M -- the same sums
    // truncated at stage M, i.e. the full suffix sums minus their values at lane M - 1 (the psi perturbation a stage sees does not
    // depend on where the horizon ends):  d x_M / d v = P3(k) X2m + zx(k) - zx(M-1),  X2m = X2(k) - X2(M-1);  d psi_M / d v = P3(k) - P3(M-1)
    T X2m = 0, Y2m = 0, zxm = 0, zym = 0, P3m = 0;
    if constexpr (SV::MSPLIT > 0) {
        constexpr int Mm = SV::MSPLIT - 1;
        X2m = X2 - s.stage_bcast(X2, Mm); Y2m = Y2 - s.stage_bcast(Y2, Mm);
        zxm = zx - s.stage_bcast(zx, Mm); zym = zy - s.stage_bcast(zy, Mm);
        P3m = P3 - s.stage_bcast(P3, Mm);
    }
    if (s.rec_writer()) {   // (four-wave kernel: all waves hold the same values, wave 0 publishes them)
        if (st) {
            T *q = s.gnb + 2 * k;
            q[0] = dt * fma(P3, X2, zx); q[GS] = dt * fma(P3, Y2, zy); q[2 * GS] = dt * P3;   // acceleration column: B = (0, 0, 0, dt)
            q[1] = fma(X2, Bdp, Bdx); q[GS + 1] = fma(Y2, Bdp, Bdy); q[2 * GS + 1] = Bdp;      // steering column
            s.gb[2 * k] = ga; s.gb[2 * k + 1] = gd;
        }
        if constexpr (SV::MSPLIT > 0) {
            if (k < SV::MSPLIT) {
                constexpr int GM = SV::GMS;
                T *q = s.gmb + 2 * k;
                q[0] = dt * fma(P3, X2m, zxm); q[GM] = dt * fma(P3, Y2m, zym); q[2 * GM] = dt * P3m;
                q[1] = fma(X2m, Bdp, Bdx); q[GM + 1] = fma(Y2m, Bdp, Bdy); q[2 * GM + 1] = Bdp;
            }
        }
        if (k <= N) {
            T *q = s.lin + LSTR * k;   // record N is all zero
            q[0] = A02; q[1] = A03; q[2] = A12; q[3] = A13; q[4] = A23; q[5] = Bdx; q[6] = Bdy; q[7] = Bdp;
            q[8] = mpp; q[9] = mpv; q[10] = mpd; q[11] = mvd; q[12] = mdd; q[13] = (T)0;
        }
    }
    xsync<SV::NTH>();
    return s.vid < n ? s.gb[s.vid] : (T)0;
}

// The second-order entries of the stage records decide between the exact and the Gauss-Newton matrix: linearize writes them only
// when the exact Hessian is wanted, and a fallback inside an iteration clears them.
template <class SV> DEV void drop_second_order_cartesian(SV &s)
{
    typedef typename SV::real T;
    T z = (T)0;
    pin(z);  // materialised here: hoisted out of the iteration loop this zero would occupy (and spill) a VGPR pair for the whole solve
    if (s.vid <= SV::N_) { T *q = s.lin + SV::LSTR * s.vid; q[8] = z; q[9] = z; q[10] = z; q[11] = z; q[12] = z; }
    xsync<SV::NTH>();
}

// stage record as the condensing recursion reads it (uniform address: one LDS broadcast per field)
template <typename T> struct Rec { T a02, a03, a12, a13, a23, bx, by, bp, mpp, mpv, mpd, mvd, mdd; };
template <class SV> DEV void load_rec(const SV &s, Rec<typename SV::real> &r, int st)
{
    const typename SV::real *q = s.lin + SV::LSTR * st;  // record N is all zero (linearize)
    r.a02 = q[0]; r.a03 = q[1]; r.a12 = q[2]; r.a13 = q[3]; r.a23 = q[4]; r.bx = q[5]; r.by = q[6]; r.bp = q[7];
    r.mpp = q[8]; r.mpv = q[9]; r.mpd = q[10]; r.mvd = q[11]; r.mdd = q[12];  // zero when the Gauss-Newton matrix is wanted
}
// Condensing in O(N^2): column j of sc * H, H = sum_s G_s^T W_s G_s + the second-order d_f rows, by an ADJOINT recursion with
// thread j = column j -- no matrix product at all:
//   start   : column j of G_N, the sensitivity of the terminal state (closed form in suffix sums of the stage Jacobians: linearize);
//   backward: p(s) = sum_{k >= s} Phi(k,s)^T W_k G_k[:,j] = W_s G_s[:,j] + A_s^T p(s+1), and with it the two rows of stage s,
//             H[2s][j] = dt p_v(s+1),  H[2s+1][j] = B_s^T p(s+1) + (mpd, mvd) . G_s[(psi, v), j]   (the m_dd diagonal: KKT assembly);
//             G_s[:,j] comes from G_{s+1}[:,j] through the exact inverse of the unit upper-triangular A_s (5 FMAs, nothing stored).
// Each thread writes its column (rows >= j) of the packed K image.  A thread whose column is born at stage j/2 carries meaningless
// (finite) values below that stage; they are never stored.  Thread 2s+1 has no row 2s: its store lands on (row n, column 2s) of the
// image -- the rhs row, which nobody reads before the factorisation writes it -- so one address serves both rows.
template <class SV> DEV void condense_adjoint(SV &s, typename SV::real sc)
{
    typedef typename SV::real T;
    constexpr int N = SV::N_, n = SV::n, GS = SV::GS, M = SV::MSPLIT, TRIPS = N - M;
    static_assert(2 * M <= N, "the upper part of the split recursion must end at stage M");
    // Split recursion (M = SV::MSPLIT > 0): the sequential depth of the recursion is halved by running stages N-1 .. M ("upper", thread j,
    // every column) and stages M-1 .. 0 ("lower", thread LOW0 + j, columns j < 2M) AT THE SAME TIME.  The lower part starts from column j
    // of G_M (closed form like G_N: linearize) and from p = 0; by linearity what its rows lack is Phi(M, s+1)^T p_j(M) seen through B_s --
    // that is G_M[:, row]^T p_j(M), a rank-4 product of two 4 x 2M tables: ONE matrix-core instruction per 16 x 16 tile, added when the KKT
    // tiles are assembled (split_fragment_a / _b below).  The upper threads leave p_j(M) in s.pm(c) and the diagonal's share in s.hm.
    bool lower = false;
    if constexpr (M > 0) lower = s.vid >= SV::LOW0 && s.vid < SV::LOW0 + 2 * M;
    if (s.vid < n || lower) {
        int j = s.vid;
        if constexpr (M > 0) j = lower ? s.vid - SV::LOW0 : s.vid;
        const T dtv = s.pt[PT_DT];
        const T Cx2 = s.cwt[0], Cy2 = s.cwt[1], Cp2 = s.cwt[2], Cv2 = s.cwt[3];
        // column j of G_N (G_M): linearize left it in the table (it depends on the linearisation only, not on the barrier weights or the shift);
        // everything downstream is linear in G, so the scaling of the objective goes in here, once
        const T *g0 = s.gnb;
        int gstr = GS;
        T wl = (T)1;
        if constexpr (M > 0) { g0 = lower ? s.gmb : s.gnb; gstr = lower ? SV::GMS : GS; wl = lower ? (T)0 : (T)1; }
        T gx = sc * g0[j], gy = sc * g0[gstr + j], gp = sc * g0[2 * gstr + j], gv = (j & 1) ? (T)0 : sc * dtv;
        T px = wl * Cx2 * gx, py = wl * Cy2 * gy, pp = wl * Cp2 * gp, pv = (T)0;   // p(N) = W_N G_N: no second-order part and no speed cost on the terminal state; lower part: 0
        T *colK = s.Lc + SV::offc_rt(j);
        int st0 = N - 1;
        if constexpr (M > 0) st0 = lower ? M - 1 : N - 1;
        Rec<T> cur;
        load_rec(s, cur, st0);
#pragma unroll 2
        for (int t = 0; t < TRIPS; ++t) {
            const int st = st0 - t;   // (lower threads of an uneven split run out of stages first: nothing is stored for st < 0)
            Rec<T> nxt;
            load_rec(s, nxt, st > 0 ? st - 1 : 0);
            const T ra = dtv * pv;
            T rd = fma(cur.bp, pp, fma(cur.by, py, cur.bx * px));
            gp = fma(-cur.a23, gv, gp);   // G_s from G_{s+1}
            gx = fma(-cur.a03, gv, fma(-cur.a02, gp, gx));
            gy = fma(-cur.a13, gv, fma(-cur.a12, gp, gy));
            const T cross = fma(cur.mvd, gv, cur.mpd * gp);
            rd += j < 2 * st ? cross : (T)0;   // G_s is exactly zero in columns 2s, 2s+1 (what the threads hold there is not)
            if (j <= 2 * st + 1) { colK[2 * st] = ra; colK[2 * st + 1] = rd; }
            // p(s) = A_s^T p(s+1) + W_s G_s  (states 1 .. N-1 carry the speed weight; p(0) is never used)
            pv = fma(cur.a23, pp, fma(cur.a13, py, fma(cur.a03, px, pv)));
            pp = fma(cur.a12, py, fma(cur.a02, px, pp));
            px = fma(Cx2, gx, px);
            py = fma(Cy2, gy, py);
            pp = fma(cur.mpv, gv, fma(Cp2 + cur.mpp, gp, pp));
            pv = fma(cur.mpv, gp, fma(Cv2, gv, pv));
            cur = nxt;
        }
        if constexpr (M > 0) {
            if (!lower && j < 2 * M) {   // p_j(M), and what the diagonal entry (j, j) lacks (max |sc H_jj| is taken before the tiles exist)
                s.pm(0)[j] = px; s.pm(1)[j] = py; s.pm(2)[j] = pp; s.pm(3)[j] = pv;
                s.hm[j] = fma(s.gmb[j], px, fma(s.gmb[SV::GMS + j], py, fma(s.gmb[2 * SV::GMS + j], pp, ((j & 1) ? (T)0 : dtv) * pv)));
            }
        }
    }
    xsync<SV::NTH>();
}
// MFMA fragments of the rank-4 product G_M^T P_M for 16-row / 16-column block t (lane: c = lane & 15 index inside the block, kk = lane >> 4
// component), zero beyond 2M.  The fourth component of G_M is dt in the acceleration columns and 0 in the steering columns.
template <class SV> DEV typename SV::real split_fragment_a(const SV &s, int t, int c, int kk)
{
    typedef typename SV::real T;
    const int i = 16 * t + c;
    const bool ok = i < 2 * SV::MSPLIT;
    const T g = s.gmb[SV::GMS * (kk < 3 ? kk : 0) + (ok ? i : 0)];
    const T v = kk == 3 ? ((i & 1) ? (T)0 : s.pt[PT_DT]) : g;
    return ok ? v : (T)0;
}
template <class SV> DEV typename SV::real split_fragment_b(const SV &s, int t, int c, int kk)
{
    typedef typename SV::real T;
    const int i = 16 * t + c;
    const bool ok = i < 2 * SV::MSPLIT;
    const T v = s.pm(kk)[ok ? i : 0];
    return ok ? v : (T)0;
}

// Strictly feasible start.  P.start = 0: a first guess of the solution inside the bounds (same rule as the CPU checker): accelerations
// approach the reference speed (time constant 1 s), steering the kinematic feed-forward of the reference's mean curvature; reference
// points 1..N only -- point 0 is a dead input (Q3).  P.start = 1: the reference's own start, every input 0 (MKZMPCPathFollower.jl:65-72),
// moved inside the first-step rate interval / the speed rows where 0 is not strictly feasible.
template <class SV> DEV bool interior_point(SV &s, typename SV::real &Uf)
{
    typedef typename SV::real T;
    constexpr int N = SV::N_;
    const T relax = s.pt[PT_RELAX], dt = s.pt[PT_DT], dtc = s.pt[PT_DTC];
    const T steer_max = s.pt[PT_STEER_MAX], a_max = s.pt[PT_A_MAX], steer_dmax = s.pt[PT_STEER_DMAX], a_dmax = s.pt[PT_A_DMAX];
    const T v_min = s.pt[PT_V_MIN], v_max = s.pt[PT_V_MAX];
    const T frac = (T)0.6, rr = s.pt[PT_RR];
    const T ffw = s.P.start == 1 ? (T)0 : (T)1;
    T len, kap;
    {
        const T rxn = __shfl_down(s.rx, 1), ryn = __shfl_down(s.ry, 1);
        const T seg = (s.lane >= 1 && s.lane < N) ? sqrt((rxn - s.rx) * (rxn - s.rx) + (ryn - s.ry) * (ryn - s.ry)) : (T)0;
        len = s.sum_stages(seg);
        kap = (s.stage_bcast(s.rp, N) - s.stage_bcast(s.rp, 1)) / fmax(len, (T)1e-6);
    }
    if constexpr (SV::MODEL_ID == 1) kap = ((s.kp0 * s.x0 + s.kp1) * s.x0 + s.kp2) * s.x0 + s.kp3;  // Frenet: curvature of the polynomial at s0
    T vref = len / ((T)(N - 1) * dt);
    if constexpr (SV::MODEL_ID == 1) vref = s.vt;
    const T sb = fmin(fmax(s.pt[PT_LB] * kap, (T)-0.9), (T)0.9);
    const T dff = ffw * fmin(fmax(atan(sb * rsqrt_((T)1 - sb * sb) / rr)  /* tan(asin(sb)) = sb / sqrt(1 - sb^2), |sb| <= 0.9 */, -frac * steer_max), frac * steer_max);
    const T aff = ffw * fmin(fmax(vref - s.v0, -frac * a_max), frac * a_max);
    T u0[2];
    // Q5: v[1] = v0 is itself bounded in the reference model -> any v0 outside the (relaxed) speed bounds is infeasible
    bool ok = s.v0 >= v_min - relax * fmax((T)1, fabs(v_min)) && s.v0 <= v_max + relax * fmax((T)1, fabs(v_max));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const T ub = j ? steer_max : a_max;
        const T d0 = (j ? steer_dmax : a_dmax) * dtc;
        const T up = s.up(j);
        T lo = fmax(-ub - relax * fmax((T)1, ub), up - d0 - relax * fmax((T)1, d0));
        T hi = fmin(ub + relax * fmax((T)1, ub), up + d0 + relax * fmax((T)1, d0));
        if (j == 0) {
            lo = fmax(lo, (v_min - relax * fmax((T)1, fabs(v_min)) - s.v0) / dt);
            hi = fmin(hi, (v_max + relax * fmax((T)1, fabs(v_max)) - s.v0) / dt);
        }
        if (!(lo < hi)) ok = false;
        const T push = (T)0.25 * (hi - lo);
        u0[j] = fmin(fmax(j ? dff : aff, lo + push), hi - push);
    }
    // Later inputs: strictly feasible BY CONSTRUCTION whenever the first input's interval above is non-empty (round 4; same rule and the argument in the CPU
    // checker's interior_point): from k = 1 on -- where the jump from a_0 is free, MKZMPCPathFollower.jl:77-79, Q1 -- the acceleration follows
    // a_k = clamp(v* - v_k, +-0.6 a_max) towards the reference speed clamped into the speed interval's margin; v approaches v* monotonically (speed rows
    // keep at least the slack of v_1) and |a_{k+1} - a_k| <= dt |a_k| <= 0.12 a_max < a_dmax dt (rate rows keep >= 0.18).  Until round 4 a speed-margin
    // override here could break a rate row: 0.2-0.5 % of wide-distribution draws started infeasible and ended as Error after one iteration.
    const T vm = fmin((T)1, (T)0.25 * (v_max - v_min));
    const T dstep = frac * steer_dmax * dt;
    const T vstar = fmin(fmax(vref, v_min + vm), v_max - vm);
    T v = s.v0 + dt * u0[0], dp = u0[1];
    Uf = s.vid == 0 ? u0[0] : (s.vid == 1 ? u0[1] : (T)0);
#pragma nounroll
    for (int k = 1; k < N; ++k) {  // uniform scalar recurrence
        const T a = ffw * fmin(fmax(vstar - v, -frac * a_max), frac * a_max);
        const T d = fmin(fmax(dff, dp - dstep), dp + dstep);
        if (s.vid == 2 * k) Uf = a;
        if (s.vid == 2 * k + 1) Uf = d;
        v += dt * a; dp = d;
    }
    return ok;
}

// ---- pieces of the blocked Cholesky both KKT back-ends share ---------------------------------------------------------------------------
// A block-step factors the 4x4 diagonal block of the current 4-column panel redundantly in every lane (4 rsqrt chains) and solves
// its panel rows against it.  What is stored is the block-LDL^T view of the same factor, K = L~ S L~^T with L~ = L D^-1 (D = blockdiag
// of the 4x4 diagonal factors, so L~ has unit diagonal blocks) and S^-1 = D^-T D^-1: the substitutions then have no dependency inside
// a block and the rhs row n comes out as S^-1 L~^-1 rhs.
template <typename T> struct Diag4 { T d00, d10, d11, d20, d21, d22, d30, d31, d32, d33; };
template <typename T> DEV Diag4<T> load_diag4(const T *pd)  // pd: the block, row-major with stride 4 (uniform addresses)
{
    Diag4<T> d;
    d.d00 = pd[0]; d.d10 = pd[4]; d.d11 = pd[5]; d.d20 = pd[8]; d.d21 = pd[9]; d.d22 = pd[10];
    d.d30 = pd[12]; d.d31 = pd[13]; d.d32 = pd[14]; d.d33 = pd[15];
    return d;
}
template <typename T> struct Chol4 {
    T r0, r1, r2, r3, l10, l20, l30, l21, l31, l32;   // reciprocal pivots and the strict lower part of the 4x4 Cholesky factor
    T i10, i20, i21, i30, i31, i32;                   // strict lower part of inv = D_j^-1 (its diagonal is r0..r3)
    bool ok;                                          // positive definite (wave-uniform: same data in every lane)
    DEV void solve_row(const T (&a)[4], T (&x)[4]) const   // x L_dd^T = a
    {
        x[0] = a[0] * r0;
        x[1] = fma(-x[0], l10, a[1]) * r1;
        x[2] = fma(-x[1], l21, fma(-x[0], l20, a[2])) * r2;
        x[3] = fma(-x[2], l32, fma(-x[1], l31, fma(-x[0], l30, a[3]))) * r3;
    }
    DEV void store_inv(T *sq) const  // D_j^-1, row-major 4x4 (the strict upper part stays zero from construction)
    {
        sq[0] = r0; sq[4] = i10; sq[5] = r1; sq[8] = i20; sq[9] = i21; sq[10] = r2; sq[12] = i30; sq[13] = i31; sq[14] = i32; sq[15] = r3;
    }
};
template <typename T> DEV Chol4<T> factor_diag4(const Diag4<T> &d)
{
    Chol4<T> c;
    c.r0 = rsqrt_(d.d00);
    c.l10 = d.d10 * c.r0; c.l20 = d.d20 * c.r0; c.l30 = d.d30 * c.r0;
    const T e11 = fma(-c.l10, c.l10, d.d11);
    c.r1 = rsqrt_(e11);
    c.l21 = fma(-c.l20, c.l10, d.d21) * c.r1; c.l31 = fma(-c.l30, c.l10, d.d31) * c.r1;
    const T e22 = fma(-c.l21, c.l21, fma(-c.l20, c.l20, d.d22));
    c.r2 = rsqrt_(e22);
    c.l32 = fma(-c.l31, c.l21, fma(-c.l30, c.l20, d.d32)) * c.r2;
    const T e33 = fma(-c.l32, c.l32, fma(-c.l31, c.l31, fma(-c.l30, c.l30, d.d33)));
    c.r3 = rsqrt_(e33);
    const T dmin = fmin(fmin(d.d00, e11), fmin(e22, e33)), dmax = fmax(fmax(d.d00, e11), fmax(e22, e33));
    c.ok = dmin > Real<T>::tiny() && dmax < (T)1e300;
    c.i10 = -c.l10 * c.r0 * c.r1;
    c.i21 = -c.l21 * c.r1 * c.r2; c.i20 = -fma(c.l21, c.i10, c.l20 * c.r0) * c.r2;
    c.i32 = -c.l32 * c.r2 * c.r3; c.i31 = -fma(c.l32, c.i21, c.l31 * c.r1) * c.r3; c.i30 = -fma(c.l32, c.i20, fma(c.l31, c.i10, c.l30 * c.r0)) * c.r3;
    return c;
}
// S^-1 y = D^-T (D^-1 y) for the 4x4 block of component j: the operands of the quad come through DPP quad_perm, all blocks at once
template <typename T> DEV T diag_solve4(const T *sinvb, T y, int j, int lane, int n)
{
    const int a = lane & 3;
    const T *blk = sinvb + 16 * ((j < n ? j : 0) >> 2);
    const T *dr = blk + 4 * a;   // row a of D^-1 (zero above the diagonal)
    const T *dc = blk + a;       // column a of D^-1 (zero above the diagonal): dc[4 m]
    const T y0 = dpp_mov0<0x00, 0xf>(y), y1 = dpp_mov0<0x55, 0xf>(y), y2 = dpp_mov0<0xaa, 0xf>(y), y3 = dpp_mov0<0xff, 0xf>(y);
    const T u = fma(dr[3], y3, dr[2] * y2) + fma(dr[1], y1, dr[0] * y0);
    const T u0 = dpp_mov0<0x00, 0xf>(u), u1 = dpp_mov0<0x55, 0xf>(u), u2 = dpp_mov0<0xaa, 0xf>(u), u3 = dpp_mov0<0xff, 0xf>(u);
    const T zz = fma(dc[12], u3, dc[8] * u2) + fma(dc[4], u1, dc[0] * u0);
    return j < n ? zz : (T)0;
}
// The structured part of K = sc*(H + input Hessian) + A^T W A + reg*I that lives on the diagonal and the (j+2, j) sub-diagonal: box and rate
// rows of A^T W A, the input-cost Hessian (MKZMPCPathFollower.jl:99-102), the (d_f, d_f) second-order entry and the shift.  Thread j < n
// computes the two entries of column j and hands them over through a 2 x n staging buffer (needs stage_form_weights done: wb = form weights).
template <class SV> DEV void kkt_diag_staging(SV &s, typename SV::real sc, typename SV::real reg, bool second_order, typename SV::real *dgs, typename SV::real *sbs)
{
    typedef typename SV::real T;
    constexpr int N = SV::N_, n = SV::n, R = SV::R;
    {   // every thread runs this (threads >= n redo column n - 1 and store the same two values): no exec-masked region here -- its join block is where
        // this toolchain's allocator twice put spill code ahead of the mask restore, at the point of highest register pressure
        const int j = s.vid < n ? s.vid : n - 1, jj = j & 1, k = j >> 1;
        const T Cu2 = s.cwt[jj ? 7 : 6], Cdl2 = s.cwt[jj ? 5 : 4];
        T dg = s.wb[j] + sc * (Cu2 + Cdl2 * (T)((k > 0) + (k < N - 1))) + reg;
        if (second_order && jj) dg += sc * s.lin[SV::LSTR * k + 12];  // m_dd of stage k: the second-order (d_f, d_f) entry
        if (j < 2) dg += s.wb[n + j];
        if (j >= 4) dg += s.wb[n + j - 2];
        const bool rate = j >= 2 && j < R;
        const T wr = rate ? s.wb[n + j] : (T)0;
        dgs[j] = dg + wr;
        sbs[j] = -wr - sc * Cdl2;
    }
    xsync<SV::NTH>();
}

// inputs of problem b through the element strides of KIO (plain arrays or packed records: kmpc_device.h)
template <class SV> DEV void load_problem_io(SV &sv, const KIO<typename SV::real> &io, int b)
{
    sv.load_problem(io.z0 + (size_t)b * io.zs, io.ref + (size_t)b * io.rs, io.vt + (size_t)b * io.vs, io.up + (size_t)b * io.us, 0);
}

// body of a solve kernel: one problem per workgroup, start order through io.perm
template <class SV> DEV void run_solver(const KP &P, const KIO<typename SV::real> &io, unsigned char *smem)
{
    if ((int)blockIdx.x >= P.B) return;
    const int b = io.perm ? io.perm[blockIdx.x] : (int)blockIdx.x;
#ifdef KMPC_POISON  // diagnostic build (make poison): every LDS word starts as NaN, so a read of a word nobody wrote shows up in the results
    typedef typename SV::real T;
    for (int e = threadIdx.x; e < SV::lds_elems(); e += SV::NTH) reinterpret_cast<T *>(smem)[e] = (T)NAN;
    __syncthreads();
#endif
    SV sv(P, smem);
    load_problem_io(sv, io, b);
    sv.solve(io, b);
}

// prologue of the KKT-pipeline diagnostics (kmpc_debug_kkt): load problem b, evaluate and linearise at the given inputs -> gradient (also in gb)
template <class SV> DEV typename SV::real debug_linearize_at(SV &sv, const KP &P, const KDbgK<typename SV::real> &io, int b, StageV<typename SV::real> &St)
{
    typedef typename SV::real T;
    sv.load_problem(io.z0, io.ref, io.vt, io.up, b);
    const T U[1] = {sv.vid < SV::n ? io.U[(size_t)b * SV::n + sv.vid] : (T)0};
    sv.eval(U, St);
    T g[1];
    sv.linearize(St, P.hessian == 1, g);
    return g[0];
}

// ---- the solve -------------------------------------------------------------------------------------------------------------------------
// One small state machine, so that every phase has a single call site (code size: a fully unrolled version was instruction-fetch bound):
//   TRIAL : Ut was just evaluated; Armijo-test it (the very first point and refactor passes skip the test)
//   after acceptance: duals, linearise, optimality test, mu, condense + factor, direction, first trial
//   REFACTOR / RESTEP : re-use the linearisation of U (larger shift / corrector dropped)
//   FINAL : last evaluation, for the predicted states, then exit
// Every decision is taken on values that are identical in all threads of the problem, so every barrier inside the hooks is reached by all.
template <class SV> DEV void solve(SV &s, const KIO<typename SV::real> &io, int b)
{
    typedef typename SV::real T;
    constexpr int NF = SV::NF, NV = SV::NV, NTH = SV::NTH;   // form / input slots per thread (slot i of a thread is index vid + NTH * i), threads per problem
    const int N = s.dim_N(), n = s.dim_n(), nf = s.dim_nf();   // compile-time constants in the compile-time-horizon back-ends, run-time values in the generic one
    const KP &P = s.P;
    const T kappa_eps = 10, kappa_mu = (T)0.2, tau_min = (T)0.99, kappa_sigma = (T)1e10, eta_phi = (T)1e-8, s_max = 100;
    // the integer options are copied out of the kernel arguments once (the argument tuple is not touched inside the loop)
    const int max_ls = P.max_ls, max_iter = P.max_iter, indef_cfg = P.indef_strategy;
    const bool warm = P.warm != 0;
    const bool exact = P.hessian == 1;
    T U[NV], Ut[NV], du[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) du[i] = (T)0;
    // slacks are iterates, advanced by s -/+ alpha * a_f^T du (as in Ipopt): recomputing b - a_f^T U would lose 7 digits to
    // cancellation once an active slack is ~1e-9
    T sup[NF], slo[NF], isu[NF], isl[NF], lu[NF], ll[NF], aut[NF], w[NF];  // isu/isl = 1/slack, refreshed when the slacks move
    bool fv[NF];
    // degenerate pairs (slack and multiplier vanish together: Newton halves them per iteration, x0.375 with the corrector; kmpc_common.h): a side of a form
    // whose slack and multiplier the last accepted full step shrank by similar shares enters K (and the recovery of its multiplier step, consistently) with
    // its barrier stiffness lambda/s scaled by KMPC_DEGEN_THETA -- the step of a double root.  Candidates are marked where the step is computed (its step-length
    // shares are the signature; two lowest mantissa bits of a_f^T du), a candidate becomes a mark when the step is accepted in full: the lowest mantissa bit of
    // the side's reciprocal slack, refreshed in that very place (one ulp of a Newton-refined reciprocal).  No register, no LDS.
    // (theta is re-derived from the bit at every use -- an integer AND, a conversion and an fma behind an opaque barrier: left to itself the compiler keeps the
    // 2 NF selected factors in registers across the iteration, which the kernels at their register limit pay with 40 more spilled registers)
    // fp64 only: the fp32 solves end at their rounding floor (error ~1e-4) before a degenerate pair's tail begins -- measured: no iteration saved, 2.5 % of
    // the launch spent on the marks -- so the fp32 instantiations compile the rule out
    constexpr bool DGR = sizeof(T) == 8;
#define DG_U(i) (DGR ? dg_theta(isu[i]) : (T)1)
#define DG_L(i) (DGR ? dg_theta(isl[i]) : (T)1)
#pragma unroll
    for (int i = 0; i < NF; ++i) { const int f = s.vid + NTH * i; fv[i] = f < nf; lu[i] = ll[i] = sup[i] = slo[i] = isu[i] = isl[i] = aut[i] = (T)0; }
    int status = 1, iters = 0, ls = 0, attempt = 0, n_polish = 0, n_accept = 0, gn_hold = 0, n_first_ok = 0;
    T *cs = s.cs;
    cs[C_ERR] = (T)1e30; cs[C_RDS] = 0; cs[C_DWL] = 0; cs[C_DWS] = 0; cs[C_HMAX] = 0; cs[C_MUF] = 0; cs[C_PHI0] = 0; cs[C_DPHI] = 0;
    cs[C_AD] = 0; cs[C_J] = 0; cs[C_LGS] = 0; cs[C_JP] = (T)1e30; cs[C_EP1] = (T)1e30; cs[C_EP2] = (T)1e30; cs[C_AL] = (T)1;
    int indef = indef_cfg == 2 ? 0 : indef_cfg, n_fail = 0;  // 2 = hybrid: GN fallback, delta_w shift from the 2nd failure on
    bool have_best = false;
    T mu = warm ? s.pt[PT_WARM_MU] : s.pt[PT_MU_INIT], sc = 1, Jt = 0, alpha = 0, reg = 0;
    bool use_exact = exact;
    enum { FIRST = 0, TRIAL = 1, REFACTOR = 2, FINAL = 3, RESTEP = 4 };
    const bool pc = P.mu_strategy == 1;
    bool corr_active = false, first_attempt = true, tiny_stop = false;
    int n_tiny = 0, n_flat = 0;
    // fp32 rounding floor (round 4, found by the out-of-distribution sweep): once the scaled error sits at a few tol the Newton steps are noise (the
    // noise-aware acceptance takes them: their predicted decrease is below the merit function's resolution) and the iterate random-walks AWAY from the
    // optimum -- error 3e-4 -> O(1) over tens of iterations, 200 iterations, IterationLimit, or a stop of the flat-objective rule on an iterate far worse
    // than an earlier one.  fp32 builds therefore keep the iterate of smallest error once that error is at the acceptable level (<= 100 tol) and stop
    // when eight iterations in a row have not improved on it: that iterate is returned, Optimal in the acceptable-level sense.  (fp64: compiled out.)
    constexpr bool FLOOR32 = sizeof(T) == 4;
    int n_stall = 0;
    T best_err = (T)1e30;
    bool final_reuse = false;  // FINAL reached with St / Jt already holding the evaluation of the returned iterate
#pragma unroll
    for (int i = 0; i < NF; ++i) s.cu(i) = s.cl(i) = (T)0;
    int mode = FIRST;
#ifdef KMPC_DRIFT_PROBE
    T drift_probe = (T)-1;
#endif
    typename SV::stage_t St;
    STAMP_DECL_AT(s)

    {
        T Uf[NV];
        const bool feas = s.interior_point(Uf);
        if (!feas) {
            status = 2;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int j = s.vid + NTH * i;
                const T ub = (j & 1) ? s.pt[PT_STEER_MAX] : s.pt[PT_A_MAX];   // (constant table indices: the generic back-end keeps the table in registers)
                U[i] = j < n ? fmin(fmax(s.up(j & 1), -ub), ub) : (T)0;
            }
            mode = FINAL;
        } else if (warm && io.warmU) {
            T dw[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) { const int j = s.vid + NTH * i; dw[i] = j < n ? io.warmU[(size_t)b * n + j] - Uf[i] : (T)0; }
            s.forms_apply(Uf, w);
            s.forms_apply(dw, aut);
            T th = 1;
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (fv[i]) {
                    T bu_, bl_;
                    s.form_bounds(s.vid + NTH * i, bu_, bl_);
                    if (aut[i] > 0) th = fmin(th, (bu_ - w[i]) / aut[i]);
                    if (aut[i] < 0) th = fmin(th, (bl_ + w[i]) / -aut[i]);
                }
            th = -s.max_any(-th) * ((T)1 - s.pt[PT_WARM_PUSH]);
#pragma unroll
            for (int i = 0; i < NV; ++i) U[i] = Uf[i] + th * dw[i];
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) U[i] = Uf[i];
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) Ut[i] = U[i];
    STAMP_AT(s, 0);
#pragma nounroll
    for (;;) {
        // the thread indices are re-materialised (opaque) at the top of every iteration: stops LICM from hoisting the index / mask / address
        // arithmetic of every phase out of the loop into registers that then live (and spill) across the whole solve
        s.refresh_ids();
        // Run-time guard of the slack iterates (round 4).  The slacks are ITERATES (advanced by alpha * a_f^T du) and the termination test trusts them.
        // Whatever makes one part from b -/+ a_f^T U -- round 3 met register-allocator spill code inside a divergent region that handed masked-off
        // lanes stale slot contents -- lets the method converge, by its own measure, on a KKT point of a SHIFTED problem.  An
        // offset, once there, stays (every later update is an increment), so ONE comparison at the end of the solve covers all of its iterates, the
        // saved best one included: the slacks of the last iterate (U at this point on every path into FINAL) against its freshly evaluated forms.
        // Drift beyond KMPC_DRIFT_TOL * max(1, |bound|, |a_f^T U|) on any live form => KMPC_NUMERICAL_ERROR, never Optimal (tolerances: kmpc_common.h).
        if (mode == FINAL && status != 2) {
            s.forms_apply(U, w);
            T drifted = (T)0;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                T bu_, bl_;
                s.form_bounds(s.vid + NTH * i, bu_, bl_);
                const T lim = (T)(sizeof(T) == 8 ? KMPC_DRIFT_TOL_F64 : KMPC_DRIFT_TOL_F32) * fmax((T)1, fmax(fmax(fabs(bu_), fabs(bl_)), fabs(w[i])));
                const bool ok_ = fabs(sup[i] - (bu_ - w[i])) <= lim && fabs(slo[i] - (bl_ + w[i])) <= lim;   // (false for NaN)
                drifted = fv[i] && !ok_ ? (T)1 : drifted;
            }
            if (s.max_any(drifted) > (T)0) { status = 3; have_best = false; best_err = (T)1e30; }
#ifdef KMPC_DRIFT_PROBE   // diagnostic build (make driftprobe): the violation output carries the largest relative drift of the last iterate's slacks instead
            drift_probe = (T)0;
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (fv[i]) {
                    T bu_, bl_;
                    s.form_bounds(s.vid + NTH * i, bu_, bl_);
                    const T sc_ = fmax((T)1, fmax(fmax(fabs(bu_), fabs(bl_)), fabs(w[i])));
                    drift_probe = fmax(drift_probe, fmax(fabs(sup[i] - (bu_ - w[i])), fabs(slo[i] - (bl_ + w[i]))) / sc_);
                }
            drift_probe = s.max_any(drift_probe);
#endif
        }
        if (mode == FINAL && (have_best || (FLOOR32 && best_err < (T)1e30)) && !tiny_stop && !(status == 0 && cs[C_ERR] <= s.pt[PT_TOL])) {
            // any later trouble (polishing noise, line-search failure, iteration cap) returns the iterate that passed (fp32: or the acceptable one of smallest error)
            s.load_best(Ut);
#pragma unroll
            for (int i = 0; i < NV; ++i) U[i] = Ut[i];
            status = 0; final_reuse = false;
        }
        // refactor / restep passes re-use the linearisation of U; a stop decided on the iterate that was just evaluated re-uses that too
        if (mode != REFACTOR && mode != RESTEP && !final_reuse) Jt = s.eval(Ut, St);
        STAMP_AT(s, 9);
        if (mode == FINAL) break;
        if (mode == TRIAL) {
            // sum of log(slack) over the forms of this thread: one log of the product (fp64 range is ample; fp32 takes one per register)
            T lgt = 0, lpr = 1;
            bool okp = true;
#pragma unroll
            // (no validity guard: the unused forms of the last register carry s = 1, ds = 0, lambda = 0, 1/s := 0 throughout, so they
            // contribute exact zeros / ones to every sum, product and maximum below -- see the start-point block)
            for (int i = 0; i < NF; ++i) {
                const T a_ = sup[i] - alpha * aut[i], b_ = slo[i] + alpha * aut[i];
                const bool pos = a_ > 0 && b_ > 0;
                okp = okp && pos;
                if (sizeof(T) == 8) lpr *= pos ? a_ * b_ : (T)1;
                else lgt += log_pos(pos ? a_ * b_ : (T)1, s.kc);
            }
            if (sizeof(T) == 8) lgt = log_pos(lpr, s.kc);
            T sm[1] = {lgt}, mx[2] = {(T)0, (T)0};
#pragma unroll
            for (int i = 0; i < NV; ++i) { mx[0] = fmax(mx[0], fabs(alpha * du[i])); mx[1] = fmax(mx[1], fabs(U[i])); }
            s.template reduce_flag<1, 2>(sm, mx, okp);
            const T slg = sm[0];
            const T phi = sc * Jt - mu * slg;
            const T phi0 = cs[C_PHI0];
            const T dphi = cs[C_DPHI], pred = -alpha * dphi;
            const bool armijo = phi - phi0 - (T)10 * Real<T>::eps() * fabs(phi0) <= -eta_phi * pred;
            const bool below_noise = dphi <= (T)0 && pred <= (T)KMPC_NOISE_ACCEPT * Real<T>::eps() * fabs(phi0);   // (kmpc_common.h)
            if (!(okp && (armijo || below_noise))) {
                // safeguard: the corrected direction is tried at the full step only; redo the step without the corrector term
                if (corr_active) {
                    mode = RESTEP;
#pragma unroll
                    for (int i = 0; i < NV; ++i) Ut[i] = U[i];
                    continue;
                }
                if (++ls >= max_ls) {
                    status = cs[C_ERR] <= s.pt[PT_TOL_X100] ? 0 : 3; mode = FINAL;   // acceptable level
#pragma unroll
                    for (int i = 0; i < NV; ++i) Ut[i] = U[i];
                    continue;
                }
                alpha *= (T)0.5;
#pragma unroll
                for (int i = 0; i < NV; ++i) Ut[i] = U[i] + alpha * du[i];
                continue;
            }
            // Ipopt's tiny-step rule: two accepted steps in a row below 10 eps relative to the iterate -> the arithmetic cannot improve
            // it; Optimal if the error is within 1e3 tol (where the rounding floor of the fp32 dual residual sits), else Error
            {
                const T stepn = mx[0], umax = fmax((T)1, mx[1]);
                n_tiny = stepn <= (T)10 * Real<T>::eps() * umax ? n_tiny + 1 : 0;
                if (n_tiny >= 2) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) U[i] = Ut[i];
                    status = cs[C_ERR] <= s.pt[PT_TOL_X1000] ? 0 : 3; tiny_stop = true; mode = FINAL; final_reuse = true; continue;
                }
            }
            // accepted: dual step from the pre-step slacks, then the slacks advance with the step
            cs[C_LGS] = slg;  // = sum log(slack) of the new iterate: the next barrier value re-uses it
            const T ad = cs[C_AD];
            cs[C_AL] = alpha;
            const bool full_step = alpha >= (T)0.9 && ad >= (T)0.9;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const T su = sup[i], sl = slo[i];
                lu[i] += ad * ((mu - s.cu(i) - lu[i] * su) * isu[i] + DG_U(i) * lu[i] * isu[i] * aut[i]);
                ll[i] += ad * ((mu - s.cl(i) - ll[i] * sl) * isl[i] - DG_L(i) * ll[i] * isl[i] * aut[i]);
                sup[i] = su - alpha * aut[i];
                slo[i] = sl + alpha * aut[i];
                // a candidate (marked where the step was computed) becomes a degenerate pair when the step was accepted (nearly) in full
                const int cb = DGR && full_step ? lsb2_get(aut[i]) : 0;
                isu[i] = fv[i] ? lsb_set(rcp_(sup[i]), (cb & 1) != 0) : (T)0; isl[i] = fv[i] ? lsb_set(rcp_(slo[i]), (cb & 2) != 0) : (T)0;
            }
#ifdef KMPC_CORRUPT_SLACK   // diagnostic build (make corrupt; tests/test_gpu_parity.py): what round 3's hazard did -- one thread's slack iterate parts from b - a_f^T U
                            // in mid-solve (consistently: its reciprocal follows) -- to show that the guard turns the resulting "Optimal" into an Error
            if (iters == 2 && s.vid == 5) { sup[0] += (T)KMPC_CORRUPT_SLACK; isu[0] = lsb_set(rcp_(sup[0]), false); }
#endif
        }
        const bool restep = mode == RESTEP;
        if (!restep) {
            if (mode != REFACTOR) {
#pragma unroll
                for (int i = 0; i < NV; ++i) U[i] = Ut[i];
                cs[C_J] = Jt;
                T g[NV];
                s.linearize(St, exact && gn_hold == 0, g);  // = use_exact of this iteration (set below, before gn_hold counts down); the gradient also stays with the back-end (grad(i))
                STAMP_AT(s, 1);
                if (mode == FIRST) {
                    s.forms_apply(U, w);
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        T bu_, bl_;
                        s.form_bounds(s.vid + NTH * i, bu_, bl_);
                        sup[i] = bu_ - w[i]; slo[i] = bl_ + w[i];
                        isu[i] = fv[i] ? lsb_set((T)1 / sup[i], false) : (T)0; isl[i] = fv[i] ? lsb_set((T)1 / slo[i], false) : (T)0;
                    }
                    T sm[1] = {(T)0}, mx[1] = {(T)0};
#pragma unroll
                    for (int i = 0; i < NV; ++i) mx[0] = fmax(mx[0], fabs(g[i]));
#pragma unroll
                    for (int i = 0; i < NF; ++i) if (fv[i]) sm[0] += log_pos(sup[i] * slo[i], s.kc);
                    s.template reduce<1, 1>(sm, mx);
                    cs[C_LGS] = sm[0];
                    const T gm = mx[0];
                    sc = gm > (T)100 ? (T)100 / gm : (T)1;  // Ipopt nlp_scaling_max_gradient
#pragma unroll
                    for (int i = 0; i < NF; ++i) { lu[i] = mu * isu[i]; ll[i] = mu * isl[i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        lu[i] = fmax(fmin(lu[i], kappa_sigma * mu * isu[i]), mu * isu[i] * ((T)1 / kappa_sigma));
                        ll[i] = fmax(fmin(ll[i], kappa_sigma * mu * isl[i]), mu * isl[i] * ((T)1 / kappa_sigma));
                    }
                }
                if (iters >= max_iter) {   // status stays ITERATION_LIMIT
                    mode = FINAL; final_reuse = true;
#pragma unroll
                    for (int i = 0; i < NV; ++i) Ut[i] = U[i];
                    continue;
                }
                ++iters;
                // optimality error (Ipopt's scaled test + unscaled duality-gap bound)
#pragma unroll
                for (int i = 0; i < NF; ++i) w[i] = lu[i] - ll[i];
                T at[NV];
                s.forms_applyT(w, at);
                T sm[2] = {(T)0, (T)0}, mx[2] = {(T)0, (T)0};
#pragma unroll
                for (int i = 0; i < NV; ++i) mx[0] = fmax(mx[0], fabs(sc * g[i] + at[i]));
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const T cu = sup[i] * lu[i], cl = slo[i] * ll[i];
                    sm[0] += lu[i] + ll[i]; sm[1] += cu + cl; mx[1] = fmax(mx[1], fmax(cu, cl));
                }
                s.template reduce<2, 2>(sm, mx);
                const T lsum = sm[0], gap = sm[1], rdm = mx[0], cm0 = mx[1];
                const T inv2nf = s.pt[PT_INV2NF];
                const T isd = s_max * rcp_(fmax(s_max, lsum * inv2nf));  // 1 / s_d
                const T err0 = fmax(rdm, cm0) * isd;
                const T tol = s.pt[PT_TOL];
                const T gap_lim = s.pt[PT_GAP_TOL] * fmax((T)1, fabs(Jt));
                cs[C_EP2] = cs[C_EP1]; cs[C_EP1] = cs[C_ERR];
                cs[C_ERR] = err0; cs[C_RDS] = rdm * isd;
                TRACE8(io.stamps, iters, err0, rdm * isd, cm0 * isd, mu, Jt, alpha, ls, (use_exact ? 1 : 0) + 2 * indef + 4 * (int)corr_active + 8 * n_tiny);
                // Ipopt's test (+ gap bound, pursued for at most 1 more iteration once Ipopt's test is met), or
                // Ipopt's "acceptable level" (error <= 100*tol for 15 iterations in a row)
                bool done = false;
                // last iterate passing Ipopt's test; fp32: or the acceptable iterate of smallest error so far (selects, not branches: the four-problem kernel
                // runs this per 16-lane row and every extra divergent region there is a place for the allocator's spill code)
                const bool better = FLOOR32 && !have_best && err0 <= s.pt[PT_TOL_X100] && err0 < best_err;
                if (FLOOR32) { n_stall = better ? 0 : n_stall + (best_err < (T)1e30 ? 1 : 0); best_err = better ? err0 : best_err; }
                if (err0 <= tol || better) s.save_best(U);
                if (err0 <= tol) have_best = true;
                if (err0 <= tol) {
                    if (gap <= gap_lim * sc || n_polish >= 1) done = true; else ++n_polish;
                } else if (n_polish > 0 && ++n_polish > 1) done = true;
                n_accept = err0 <= s.pt[PT_TOL_X100] ? n_accept + 1 : 0;
                // rounding floor: the objective has not moved by more than 20 eps |J| for 12 iterations in a row -> the arithmetic cannot
                // improve the iterate (fp32, large costs: the dual residual never settles below 100 tol); Optimal within 1e3 tol
                n_flat = fabs(Jt - cs[C_JP]) <= (T)20 * Real<T>::eps() * fmax((T)1, fabs(Jt)) ? n_flat + 1 : 0;
                cs[C_JP] = Jt;
                if (n_flat >= 12 && err0 <= s.pt[PT_TOL_X1000]) done = true;
                if (FLOOR32 && !have_best && n_stall >= 8) done = true;
                if (done || n_accept >= 15) {
                    status = 0; mode = FINAL; final_reuse = true;
#pragma unroll
                    for (int i = 0; i < NV; ++i) Ut[i] = U[i];
                    continue;
                }
                const T mu_min = fmax(s.pt[PT_TOL_D100], fmin(s.pt[PT_TOL_D10], (T)0.1 * gap_lim * sc * inv2nf));
                cs[C_MUF] = mu_min;
#pragma nounroll
                for (; !pc;) {  // monotone barrier update (mu_strategy 0)
                    T dm[1] = {(T)0}, cm[1] = {(T)0};
#pragma unroll
                    for (int i = 0; i < NF; ++i)
                        if (fv[i]) cm[0] = fmax(cm[0], fmax(fabs(sup[i] * lu[i] - mu), fabs(slo[i] * ll[i] - mu)));
                    s.template reduce<0, 1>(dm, cm);
                    if (fmax(rdm, cm[0]) * isd <= kappa_eps * mu && mu > mu_min) mu = fmax(mu_min, fmin(kappa_mu * mu, mu * sqrt(mu)));
                    else break;
                }
                use_exact = exact && gn_hold == 0; reg = 0; attempt = 0;  // GN is held for 2 iterations after an indefinite exact Hessian
                if (gn_hold > 0) --gn_hold;
                // in shift mode the previous iteration's delta_w / 3 is the first trial (dropped below 1e-9 * max|H_jj|)
                // ... and after two first-trial successes in a row the unshifted matrix is tried first again: a decaying shift slows the end game of the solves
                // that have left the non-convex region
                if (use_exact && indef == 1 && cs[C_DWS] > (T)0) { reg = cs[C_DWS] / (T)3; if (reg < (T)1e-9 * cs[C_HMAX] || n_first_ok >= 2) reg = 0; }
                first_attempt = true;
                STAMP_AT(s, 2);
            }
            // K = sc*H + A^T Sigma A with the affine right-hand side -sc*g riding along
#pragma unroll
            for (int i = 0; i < NF; ++i) w[i] = DG_U(i) * lu[i] * isu[i] + DG_L(i) * ll[i] * isl[i];
            s.stage_form_weights(w);
            STAMP_AT(s, 6);
            const bool factored = s.kkt_factor(sc, reg, use_exact && indef == 1 && first_attempt);
            first_attempt = false;
            STAMP_AT(s, 5);
            if (!factored) {
                // Indefinite exact Hessian: strategy 0 -> Gauss-Newton for this and the next 2 iterations; 1 -> Ipopt's inertia
                // correction K + delta_w*I, delta_w = 1e-2*max|sc*H_jj| (x10) the first time, last/3 (x3: back to the last shift that worked) afterwards;
                // 2 -> 0 until the second failure, 1 from then on (Gauss-Newton leaves a saddle only slowly)
                bool giveup = ++attempt >= 40;
                if (!giveup) {
                    if (use_exact && indef == 1) {
                        const T hmax = cs[C_HMAX], dw_last = cs[C_DWL];
                        if (reg == (T)0) reg = dw_last > (T)0 ? fmax((T)1e-10 * hmax, dw_last / (T)3) : (T)1e-2 * hmax;
                        else reg *= dw_last > (T)0 ? (T)KMPC_DW_GROW : (T)10;
                        if (reg > (T)1e2 * hmax) { use_exact = false; reg = 0; s.drop_second_order(); }
                    } else if (use_exact) {
                        use_exact = false; gn_hold = 2; s.drop_second_order();
                        // (hybrid: shift mode from the first failure up to N = 28 on cold starts, from the second beyond and on warm starts -- pooled
                        // worst-of-4096 statistics; a warm start from a poor point begins at mu = 1e-6, where shift mode right away stalls)
                        if (indef_cfg == 2 && ++n_fail >= ((N >= 32 || warm) ? 2 : 1)) { indef = 1; gn_hold = 0; }
                    } else reg = reg == (T)0 ? (T)1e-8 : reg * (T)100;  // last resort: shift the Gauss-Newton matrix
                    mode = REFACTOR;
                } else { status = 3; mode = FINAL; final_reuse = true; }
#pragma unroll
                for (int i = 0; i < NV; ++i) Ut[i] = U[i];
                continue;
            }
            if (use_exact && reg > (T)0) cs[C_DWL] = reg;
            if (use_exact) { cs[C_DWS] = reg; n_first_ok = attempt == 0 ? n_first_ok + 1 : 0; }
#pragma unroll
            for (int i = 0; i < NF; ++i) s.cu(i) = s.cl(i) = (T)0;
            corr_active = false;
            if (pc) {
                // Mehrotra predictor: affine-scaling step on the same factor -> this iteration's barrier target
                T dua[NV];
                s.kkt_affine(dua);
                s.forms_apply(dua, aut);
                // step lengths to the boundary as reciprocals: 1/alpha = max(1, max_f(-ds/s)); for the affine step -dlam/lam = 1 + ds/s
                T sm[1] = {(T)0}, mx[2] = {(T)1, (T)1};
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const T qu = aut[i] * isu[i], ql = aut[i] * isl[i];  // -ds_u/s_u, ds_l/s_l
                    mx[0] = fmax(mx[0], fmax(qu, -ql));
                    mx[1] = fmax(mx[1], fmax((T)1 - DG_U(i) * qu, (T)1 + DG_L(i) * ql));
                    sm[0] += sup[i] * lu[i] + slo[i] * ll[i];
                }
                s.template reduce<1, 2>(sm, mx);
                const T apa = rcp_(mx[0]), ada = rcp_(mx[1]);
                T sa[1] = {(T)0}, dm[1] = {(T)0};
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const T su = sup[i], sl = slo[i], dsu = -aut[i], dsl = aut[i];
                    const T dlu = -lu[i] - DG_U(i) * lu[i] * isu[i] * dsu, dll = -ll[i] - DG_L(i) * ll[i] * isl[i] * dsl;
                    sa[0] += (su + apa * dsu) * (lu[i] + ada * dlu) + (sl + apa * dsl) * (ll[i] + ada * dll);
                    s.cu(i) = dsu * dlu; s.cl(i) = dsl * dll;
                }
                s.template reduce<1, 0>(sa, dm);
                const T mucur = sm[0] * s.pt[PT_INV2NF], muaff = sa[0] * s.pt[PT_INV2NF];
                const T r3 = muaff * rcp_(mucur);
                mu = fmax(cs[C_MUF], fmin((T)1, r3 * r3 * r3) * mucur);
                // no barrier target far below the dual infeasibility: 1e-3 of it, 1e-2 in shifted (non-convex) iterations, where a small mu
                // makes the damped steps hug the bounds (fraction-to-the-boundary steps of 1e-4 ... 1e-8)
                // (the floor of the shifted case comes from the LDS table: written as a select between two fp64 literals it was mis-evaluated in the
                // one instantiation that spills to AGPRs -- Frenet functor, N = 28, fp64 -- which the iteration-count parity test caught)
                // ... but not while the solve is visibly converging: outside shift mode the floor is dropped whenever the optimality error fell in each of the
                // last two iterations -- there it only slowed the end game (mean iterations 7.94 -> 7.45 at N = 20, 9.83 -> 8.97 at N = 50, 6.93 -> 6.48 at
                // N = 8; worst-of-4096 statistics unchanged).  With no floor at all outside shift mode a rare problem cycles to the iteration cap.
                const bool shifted = use_exact && reg > (T)0;
                const bool converging = cs[C_ERR] < cs[C_EP1] && cs[C_EP1] < cs[C_EP2];
                // ... and after a tiny fraction-to-the-boundary step (alpha < KMPC_UNSTICK) the floor applies without its cap at the current mean complementarity:
                // a warm start from a wrong point begins with mu = 1e-7 and slacks 1e-5 off the bounds -- mu_cur can then never grow, the Newton direction
                // keeps pointing into the bound and the solve crawls to the iteration cap in steps of 1e-6 (1 of 32 768 wrong-point warm starts at N = 8 and
                // at N = 20, tools/warm_probe.py; with the rule: at most 20 / 34 iterations; warm starts from the own solution and cold starts unchanged)
                const bool stuck = cs[C_AL] < (T)KMPC_UNSTICK;
                const T floor_k = shifted ? s.pt[PT_IKRD_NC] : ((stuck || indef == 1 || !converging) ? (T)KMPC_IKRD : (T)0);
                mu = fmax(mu, fmin(stuck ? (T)1e30 : mucur, cs[C_RDS] * floor_k));
                corr_active = true;
                STAMP_AT(s, 7);
            }
        } else {  // RESTEP: same factor, corrector term dropped
#pragma unroll
            for (int i = 0; i < NF; ++i) s.cu(i) = s.cl(i) = (T)0;
            corr_active = false;
        }
        // centering (+ corrector) part of the step: du = K^{-1}(-sc*g - A^T((mu - corr)/s_u - (mu - corr)/s_l))
#pragma unroll
        for (int i = 0; i < NF; ++i) w[i] = -((mu - s.cu(i)) * isu[i] - (mu - s.cl(i)) * isl[i]);
        {
            T bw[NV];
            s.forms_applyT(w, bw);
            s.kkt_direction(bw, du);
        }
        STAMP_AT(s, 15);
        s.forms_apply(du, aut);
        const T tau = fmax(tau_min, (T)1 - mu);
        {
            T sm[1] = {(T)0}, mx[2] = {(T)0, (T)0};
#pragma unroll
            for (int i = 0; i < NV; ++i) sm[0] += s.vid + NTH * i < n ? sc * s.grad(i) * du[i] : (T)0;
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (fv[i]) {
                    const T su = sup[i], sl = slo[i], dsu = -aut[i], dsl = aut[i];
                    const T dlu = (mu - s.cu(i) - lu[i] * su) * isu[i] - DG_U(i) * lu[i] * isu[i] * dsu;
                    const T dll = (mu - s.cl(i) - ll[i] * sl) * isl[i] - DG_L(i) * ll[i] * isl[i] * dsl;
                    sm[0] += mu * (isu[i] - isl[i]) * aut[i];
                    // shares of the slack and of its multiplier that the full step takes off (the step lengths' inputs) -- and the signature of a degenerate
                    // pair: both above 0.3 and within 0.2 of each other.  The two candidate marks ride in the two lowest mantissa bits of a_f^T du until
                    // the step is accepted (no register)
                    const T qsu = -dsu * isu[i], qsl = -dsl * isl[i], qlu = -dlu * rcp_(lu[i]), qll = -dll * rcp_(ll[i]);
                    mx[0] = fmax(mx[0], fmax(qsu, qsl));
                    mx[1] = fmax(mx[1], fmax(qlu, qll));
                    const bool cu_ = qsu > (T)0.3 && qlu > (T)0.3 && fabs(qsu - qlu) < (T)0.2, cl_ = qsl > (T)0.3 && qll > (T)0.3 && fabs(qsl - qll) < (T)0.2;
                    if (DGR) aut[i] = lsb2_set(aut[i], (cu_ ? 1 : 0) | (cl_ ? 2 : 0));
                }
            s.template reduce<1, 2>(sm, mx);
            // fraction to the boundary: alpha = min(1, tau * min(-s/ds)) = tau / max(tau, max(-ds/s))
            alpha = tau * rcp_(fmax(tau, mx[0]));
            cs[C_AD] = tau * rcp_(fmax(tau, mx[1]));
            cs[C_PHI0] = sc * cs[C_J] - mu * cs[C_LGS];
            cs[C_DPHI] = sm[0];  // d/dalpha of phi_mu: (sc*g + A^T(mu/s_u - mu/s_l))^T du
            TRACE8(io.stamps, 128 + (iters & 127), reg, cs[C_HMAX], cs[C_DWL], cs[C_DWS], attempt, mx[0], sm[0], mx[1]);
        }
        ls = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) Ut[i] = U[i] + alpha * du[i];
        mode = TRIAL;
        STAMP_AT(s, 8);
    }
    STAMP_AT(s, 10);
    // ---- outputs (St / Jt are the evaluation of the returned U) ------------------------------------
    s.forms_apply(U, w);
    T viol = -(T)1e30;
#pragma unroll
    for (int i = 0; i < NF; ++i)
        if (fv[i]) {
            const int f = s.vid + NTH * i;
            T bu_, bl_;
            s.form_bounds(f, bu_, bl_);
            viol = fmax(viol, fmax(w[i] - (bu_ - s.form_relax(f, true)), -w[i] - (bl_ - s.form_relax(f, false))));
        }
    viol = s.max_any(viol);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = s.vid + NTH * i;
        if (j < n) {
            if (io.outU) io.outU[(size_t)b * n + j] = U[i];
            if (io.warmU) io.warmU[(size_t)b * n + j] = U[i];
            if (j < 2) io.u0[(size_t)b * io.u0s + j] = U[i];
        }
    }
    if (io.outX && s.vid <= N) {
        T *o = io.outX + ((size_t)b * (N + 1) + s.vid) * 4;
        o[0] = St.x + s.xoff(); o[1] = St.y + s.yoff(); o[2] = St.psi; o[3] = St.v;
    }
    STAMP_AT(s, 11);
    STAMP_OUT_AT(s, io.stamps, b);
    if (s.vid == 0) {
#ifdef KMPC_DRIFT_PROBE
        viol = drift_probe;
#endif
        io.status[(size_t)b * io.is] = status;
        if (io.cost) io.cost[(size_t)b * io.ss] = Jt;
        if (io.viol) io.viol[(size_t)b * io.ss] = viol;
        if (io.iters) io.iters[(size_t)b * io.is] = iters;
    }
    // small-batch host entry point (kmpc_solve_batch_host, B <= 16): the outputs above went to pinned host memory; the host spins on this counter instead
    // of going through the runtime's completion path (launch + synchronisation of an EMPTY kernel cost 18 us of a 53 us warm solve)
    if (io.done) {
        __threadfence_system();
        xsync<NTH>();
        if (s.vid == 0) __hip_atomic_fetch_add(io.done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace ipm
